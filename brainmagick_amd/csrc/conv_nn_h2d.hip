// The round-2..5 main loop of the wide f16x2 conv, kept for A/B runs (BM_CONV_LDSDMA=1): weight slab of stage s + 2 by
// LDS DMA into three slab buffers, input window of the next chunk through registers, raw s_barrier with hand-counted
// vmcnt once per STAGE (the slab is shared by the workgroup).  The production main loop is conv_nn_h2w.hip.
// LDS: A [3 buffers][2 planes][2 groups][64 MW rows], X [2 buffers][2 planes][2 groups][256 columns] x 16 B.
#include "conv_h2_common.h"

template <int N, int MW>
__device__ __forceinline__ void h2_barrier(f16x8 (&ah)[MW], f16x8 (&bh)[3]) {
    if constexpr (MW == 5)
        asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2])
                     : "n"(N) : "memory");
    else if constexpr (MW == 4)
        asm volatile("s_waitcnt vmcnt(%7) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2])
                     : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%5) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2])
                     : "n"(N) : "memory");
}


#ifdef HG_TRACE
__device__ unsigned ch_trace_buf[64 * 4 * 24];
extern "C" int bm_debug_trace_read_conv_h2d(unsigned* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_trace_buf), sizeof(unsigned) * 64 * 4 * 24);
}
#endif

template <int KS, int MW>
__global__ __launch_bounds__(256, 1) void conv_nn_h2d_kernel(ConvH2Args args) {
    const ConvNNArgs& a = args.c;
#ifdef HG_TRACE
    const unsigned t_kernel0 = (unsigned)__builtin_readcyclecounter();
#endif
    constexpr int NW = 3;
    constexpr int HBM = 64 * MW;                      // rows of the workgroup tile
    constexpr int HASLAB = 2 * 2 * HBM;               // 16-byte slots of one A buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);       // [3 buffers][2 planes][2 groups][HBM]
    u32x4* Xs = As + 3 * HASLAB;                      // [2 buffers][2 planes][2 groups][HXWP]
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
    const int m0 = mtile * HBM;
    const int n0 = ntile * HBN;
    const int halo = (KS >> 1) * a.dil;
    const int XW = HBN + 2 * halo;                    // <= 224

    const int g = a.widx ? a.widx[b] : 0;
    const int nchunk32 = a.nchunk;
    const int n16 = (a.Cin + 15) >> 4;                // channels past Cin read 0 through the bounds check
    const int nstage = n16 * KS;
    // packed weights in 16-byte slots: [g][chunk32][tap][plane][4][Mpad]
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (long)g * nchunk32 * KS * 8 * a.Mpad + m0 + lane;
    float sx, sx_inv;
    h2_scale_from_amax(bm_amax_load(args.x_amax), sx, sx_inv);

    // input window of this segment through a bounds-checked buffer descriptor: channels past Cin read 0
    const unsigned long long xaddr = (unsigned long long)(a.x + (long)b * a.x_bstride);
    i32x4h xr;
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
    f16x8 af[2][MW], bf[2][NW];                       // operand fragments [plane 0 = hi, 1 = lo]

    // DMA of the A slab of stage S_ (clamped to the last stage) into A buffer BUF3_: 4 MW pieces of 64 slots,
    // MW per wavefront
#define DMA_A(S_, BUF3_)                                                                          \
    {                                                                                             \
        const int sc = (S_) < nstage ? (S_) : nstage - 1;                                         \
        const int c16 = sc / KS, jj = sc - c16 * KS;                                              \
        const u32x4* src = wg + ((long)((c16 >> 1) * KS + jj) * 8 + (c16 & 1) * 2) * a.Mpad;      \
        _Pragma("unroll") for (int i = 0; i < MW; ++i) {                                          \
            const int k = wave + 4 * i;                                                           \
            const int run = k / MW, rb = k - run * MW;          /* run = plane * 2 + group */     \
            const int plane = run >> 1, kg = run & 1;                                             \
            __builtin_amdgcn_global_load_lds(                                                     \
                (const void*)(src + (long)(plane * 4 + kg) * a.Mpad + rb * 64),                   \
                (__attribute__((address_space(3))) void*)(As + (BUF3_) * HASLAB + run * HBM + rb * 64), 16, 0, 0); \
        }                                                                                         \
    }
    // 16 channels x 1 column of the input window of chunk C16_ (zeros past the last chunk: offset out of range)
#define LOAD_X(C16_)                                                                              \
    {                                                                                             \
        const int cb = (C16_) * 16 * crow + xoff0;                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) xreg[r] = ch_ld32(xr, cb + r * crow);      \
    }
#define CH_WAIT_X(N_)                                                                             \
    asm volatile("s_waitcnt vmcnt(%16)"                                                           \
                 : "+v"(xreg[0]), "+v"(xreg[1]), "+v"(xreg[2]), "+v"(xreg[3]), "+v"(xreg[4]), "+v"(xreg[5]),  \
                   "+v"(xreg[6]), "+v"(xreg[7]), "+v"(xreg[8]), "+v"(xreg[9]), "+v"(xreg[10]), "+v"(xreg[11]), \
                   "+v"(xreg[12]), "+v"(xreg[13]), "+v"(xreg[14]), "+v"(xreg[15])                 \
                 : "n"(N_) : "memory");
#define STORE_X(BUF_)                                                                             \
    {                                                                                             \
        u32x4* xd = Xs + (BUF_) * HXSLAB + tid;                                                   \
        _Pragma("unroll") for (int kg = 0; kg < 2; ++kg) {                                        \
            u32x4 hi, lo;                                                                         \
            split8h(xreg + 8 * kg, sx, hi, lo);                                                   \
            xd[(0 * 2 + kg) * HXWP] = hi;                                                         \
            xd[(1 * 2 + kg) * HXWP] = lo;                                                         \
        }                                                                                         \
    }
    // single fragments of the stage in A buffer ABUF_ / X buffer XB_, tap J_ (plane 0 = hi, 1 = lo)
#define FRAG_A(P_, ABUF_, MT_)                                                                    \
    af[P_][MT_] = __builtin_bit_cast(f16x8, As[(ABUF_) * HASLAB + ((P_) * 2 + h) * HBM + wm * (MW * 32) + nl + (MT_) * 32]);
#define FRAG_B(P_, XB_, J_, NT_)                                                                  \
    bf[P_][NT_] = __builtin_bit_cast(f16x8, Xs[(XB_) * HXSLAB + ((P_) * 2 + h) * HXWP + wn * (NW * 32) + nl + (J_) * a.dil + (NT_) * 32]);
    // "slab s + 1 landed" + workgroup barrier: N_ younger VMEM instructions may stay in flight.  The statement
    // names the A.hi and B.hi fragments as read-write operands: the compiler moves register-only MFMAs freely
    // across an asm statement ("memory" does not order them), and this pins the readers of B.hi before it.
#define CH_BARRIER(N_) h2_barrier<N_, MW>(af[0], bf[0]);
#define TERM(PA_, PB_)                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        _Pragma("unroll") for (int nt = 0; nt < NW; ++nt)                                         \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA_][mt], bf[PB_][nt], acc[mt][nt], 0, 0, 0);

    // prologue: slabs 0 and 1, input window of chunk 0
    DMA_A(0, 0)
    DMA_A(1, 1)
    LOAD_X(0)
    CH_WAIT_X(0)
    STORE_X(0)
    if (KS == 1) LOAD_X(1)          // 1x1 convs: the window of chunk c + 2 is requested in stage c (see below)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // Stage s = (16-channel chunk c16, tap j), A buffer s % 3, X buffer c16 & 1; 3 MW NW MFMAs in three terms:
    // hi*lo of the PREVIOUS stage (operands still in registers; zeros before the first stage), lo*hi, hi*hi, then
    // the barrier.  One slot per MFMA, in source order (scheduling fence after each): a lone wavefront per SIMD
    // issues one instruction every 8 cycles, so an MFMA (32 cycles) hides about 3 more, and anything issued in a
    // burst -- 16 window loads, a read that is waited for at once -- leaves the matrix pipe idle (cycle trace:
    // scripts/trace_conv.py).  Behind the MFMAs of a stage, at most a few instructions each:
    //   term 0: the stage's first fragments (B.hi, A.lo);
    //   term 1: the MW DMA copies of slab s + 2, the A.hi / B.lo fragments (term 2 / the next stage's term 0), and
    //     3 taps, j == 0:      the 16 loads of the input window of chunk c16 + 1;
    //     3 taps, j == KS - 1: that window, split pair by pair into X buffer (c16 + 1) & 1;
    //     1x1:                 window c16 + 1 (requested in stage c16 - 1) split pair by pair, each pair's
    //                          registers refilled with window c16 + 2 right behind.
    // VMEM queue order of a stage: slab s + 2 (MW copies), then the window loads.  Counted waits: the window is
    // consumed behind the copies of its stage (3 taps: younger are the copies of stages j = 1, 2; 1x1: this
    // stage's copies); at the barrier slab s + 1 must have landed: younger are this stage's copies and the window
    // loads issued since (3 taps: j = 0 and j = 1; 1x1: the window just waited for is older than nothing needed).
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) af[0][mt] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int nt = 0; nt < NW; ++nt) bf[1][nt] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int TN = MW * NW;
    constexpr int WS0 = TN + MW, WSN = 2 * TN - MW;   // slots behind the copies: [WS0, WS0 + WSN)
    unsigned ph[4], pw[4];                            // one 8-channel group of the window being split
    int s = 0;
    int ab3 = 0;                                      // s % 3
#ifdef HG_TRACE
    unsigned tr[8], tacc[3][8];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) tacc[i][k] = 0;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned t_loop0 = (unsigned)__builtin_readcyclecounter();
#endif
    for (int c16 = 0; c16 < n16; ++c16) {
        const int xbuf = c16 & 1;
        h2_static_for<KS>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            CH_T(0)
            const int ab_next = ab3 == 2 ? 0 : ab3 + 1;
            const int ab_next2 = ab_next == 2 ? 0 : ab_next + 1;
            // source of slab s + 2 (clamped to the last stage) and of the window to request
            const int sc = s + 2 < nstage ? s + 2 : nstage - 1;
            const int dc16 = sc / KS, djj = sc - dc16 * KS;
            const u32x4* dsrc = wg + ((long)((dc16 >> 1) * KS + djj) * 8 + (dc16 & 1) * 2) * a.Mpad;
            const int cb = (c16 + (KS == 1 ? 2 : 1)) * 16 * crow + xoff0;
            u32x4* xd = Xs + (xbuf ^ 1) * HXSLAB + tid;
            h2_static_for<3 * TN>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                constexpr int term = n / TN, w = n % TN, mt = w / NW, nt = w % NW;
                constexpr int pa = term == 1 ? 1 : 0, pb = term == 0 ? 1 : 0;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[pa][mt], bf[pb][nt], acc[mt][nt], 0, 0, 0);
                if constexpr (term == 0) {                                 // the stage's first fragments
                    if constexpr (w < NW) FRAG_B(0, xbuf, j, w)
                    else if constexpr (w < NW + MW) FRAG_A(1, ab3, w - NW)
                }
                // A.hi / B.lo (free once term 0 is issued) behind the copies: every wait on LDS data is an
                // lgkmcnt(0), so no read may be issued right in front of the first MFMA of a term
                if constexpr (n >= TN + MW && n < TN + 2 * MW) FRAG_A(0, ab3, n - TN - MW)
                if constexpr (n >= TN + 2 * MW && n < TN + 2 * MW + NW) FRAG_B(1, xbuf, j, n - TN - 2 * MW)
                if constexpr (term == 1 && w < MW) {                       // DMA piece w of slab s + 2
                    constexpr int k4 = w;
                    const int k = wave + 4 * k4;
                    const int run = k / MW, rb = k - run * MW;             // run = plane * 2 + group
                    __builtin_amdgcn_global_load_lds(
                        (const void*)(dsrc + (long)((run >> 1) * 4 + (run & 1)) * a.Mpad + rb * 64),
                        (__attribute__((address_space(3))) void*)(As + ab_next2 * HASLAB + run * HBM + rb * 64), 16, 0, 0);
                }
                if constexpr (n >= WS0) {
                    constexpr int q = n - WS0;
                    if constexpr (KS != 1 && j == 0) {                     // window loads, 16 over WSN slots
                        constexpr int r0 = (q * 16 + WSN - 1) / WSN, r1 = ((q + 1) * 16 + WSN - 1) / WSN;
                        h2_static_for<r1 - r0>([&](auto rc) __attribute__((always_inline)) {
                            constexpr int r = r0 + decltype(rc)::value;
                            if constexpr (r < 16) xreg[r] = ch_ld32(xr, cb + r * crow);
                        });
                    }
                    if constexpr (KS == 1 || j == KS - 1) {                // split units, 8 over WSN slots
                        if constexpr (q == 0) {
                            if (KS == 1) CH_WAIT_X(MW) else CH_WAIT_X(2 * MW)
                        }
                        constexpr int u0 = (q * 8 + WSN - 1) / WSN, u1 = ((q + 1) * 8 + WSN - 1) / WSN;
                        h2_static_for<u1 - u0>([&](auto uc) __attribute__((always_inline)) {
                            constexpr int u = u0 + decltype(uc)::value;
                            if constexpr (u < 8) {
                                // the group of pairs 0-3 is written one unit late: not in front of term 2's wait
                                if constexpr (u == 4) {
                                    xd[(0 * 2 + 0) * HXWP] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                                    xd[(1 * 2 + 0) * HXWP] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                                }
                                ch_split_pair(xreg[2 * u], xreg[2 * u + 1], sx, ph[u & 3], pw[u & 3]);
                                if constexpr (u == 7) {
                                    xd[(0 * 2 + 1) * HXWP] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                                    xd[(1 * 2 + 1) * HXWP] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                                }
                                if constexpr (KS == 1) {
                                    xreg[2 * u] = ch_ld32(xr, cb + (2 * u) * crow);
                                    xreg[2 * u + 1] = ch_ld32(xr, cb + (2 * u + 1) * crow);
                                }
                            }
                        });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            CH_T(4)
            if (KS == 1) CH_BARRIER(16 + MW) else if (j == KS - 1) CH_BARRIER(MW) else CH_BARRIER(16 + MW)
            CH_T(5)
#ifdef HG_TRACE
            tacc[j][0] += tr[4] - tr[0]; tacc[j][4] += tr[5] - tr[4]; tacc[j][7] += 1;
#endif
            ab3 = ab_next;
            ++s;
        });
    }
#ifdef HG_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const unsigned t_loop1 = (unsigned)__builtin_readcyclecounter();
    if (blockIdx.x < 64 && lane == 0)
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) ch_trace_buf[((blockIdx.x * 4 + wave) * 3 + i) * 8 + k] = tacc[i][k];
#endif
    TERM(0, 1)                                        // last stage
    // drain the (clamped, unused) copies of the last stages before the LDS is re-used by the epilogue
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#undef DMA_A
#undef LOAD_X
#undef CH_WAIT_X
#undef STORE_X
#undef FRAG_A
#undef FRAG_B
#undef TERM
#undef CH_BARRIER

#ifdef HG_TRACE
    h2_tile_epilogue<MW>(args, acc, smem, b, g, m0, n0, ntile, sx_inv, tid, lane, wm, wn, nl, h, ch_trace_buf);
#else
    h2_tile_epilogue<MW>(args, acc, smem, b, g, m0, n0, ntile, sx_inv, tid, lane, wm, wn, nl, h);
#endif
#ifdef HG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tile's stores have left the CU
    if (blockIdx.x < 64 && lane == 0) {
        unsigned* o = ch_trace_buf + ((blockIdx.x * 4 + wave) * 3) * 8;
        o[5] = t_loop0 - t_kernel0;
        o[6] = (unsigned)__builtin_readcyclecounter() - t_loop1;
        o[8 + 5] = t_loop1 - t_loop0;
    }
#endif
}


template <int KS, int MW>
static int launch_h2d(const ConvH2Args& args, size_t lds, unsigned nblocks, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_h2d_kernel<KS, MW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "conv_nn_h2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_nn_h2d_kernel<KS, MW>), dim3(nblocks), dim3(256), lds, stream, args);
    return BM_OK;
}

int bm_launch_conv_nn_h2d(const ConvH2Args& args, int KS, int mw, size_t lds, unsigned nblocks, hipStream_t stream) {
#define H2D_DISPATCH(MW_) return KS == 1 ? launch_h2d<1, MW_>(args, lds, nblocks, stream) : launch_h2d<3, MW_>(args, lds, nblocks, stream);
    switch (mw) {
        case 5: H2D_DISPATCH(5)
        case 4: H2D_DISPATCH(4)
        default: H2D_DISPATCH(2)
    }
#undef H2D_DISPATCH
}
