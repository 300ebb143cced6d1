// fp32-ACCURATE implicit-GEMM conv on the f16 matrix cores, compute mode "f16x2".
//
// Every fp32 operand x is scaled by a power of two s (so that the largest magnitude of its tensor -- or, for
// the weights, of its output row -- lands in [2^14, 2^15)) and split into TWO f16 numbers
//     x * s = hi + lo + e,   hi = f16(x s),  lo = f16(x s - hi),   |e| <= 2^-22 |x s|
// (11 + 11 significand bits, round to nearest twice).  A product a * b is evaluated as the three partial
// products  a.lo * b.hi + a.hi * b.lo + a.hi * b.hi  (the dropped a.lo * b.lo is <= 2^-22 |a b|), each an
// EXACT f16 x f16 product accumulated in fp32 by v_mfma_f32_32x32x16_f16; the accumulator is multiplied by
// the (exact) inverse scales in the epilogue.  Three 16-bit MFMAs per fp32-accurate 32x32x16 block instead of
// the six of the 3 x bf16 split (conv_nn_x3w.hip): half the matrix-core time for the same parity tolerances --
// measured rel-L2 against fp64 is that of an fp32 FMA chain (tests/test_exact_f32_gpu.py), because the fp32
// accumulation error dominates both.  The error bound is norm-wise (relative to the largest element of the
// tensor / weight row), not element-wise: an element 2^-18 below its tensor's maximum keeps fewer than 22
// bits; the reference's activations (BatchNorm'ed / clamped, bm/norm.py:332-333) span a few binades.
//
// Structure = conv_nn_x3w.hip (ONE workgroup of four wavefronts per CU, one wavefront per SIMD, wavefront
// tile (32 MW) x 96 as MW x 3 MFMA accumulators, workgroup tile (64 MW) x 192, MW in {5, 4, 2}; weight slab
// of stage s + 2 by LDS DMA, input window of the next chunk through registers, raw s_barrier with hand-counted
// vmcnt), with two operand planes: stage = (16-channel chunk, tap) = 15 MW MFMAs per wavefront.
// LDS: A [3 buffers][2 planes][2 groups][64 MW rows], X [2 buffers][2 planes][2 groups][256 columns] x 16 B.
// Packed weights: [g][chunk32][tap][plane][4 groups][Mpad] 16-byte slots (8 f16 channels), followed by the
// per-row inverse scales [G][Mpad] fp32 (bm_pack_weights_h2).
#include <cstdlib>
#include <cstring>
#include <utility>
#include "conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4h __attribute__((ext_vector_type(4)));

#define HBN 192           // columns of the workgroup tile (2 wavefront columns x 3 blocks)
#define HXWP 256          // padded x-window width (>= 192 + 2 * 16; one column per thread)
#define HXSLAB (2 * 2 * HXWP)     // 16-byte slots of one X buffer

struct ConvH2Args {
    ConvNNArgs c;
    const float* x_amax;      // [BM_AMAX_SHARDS] shards whose maximum bounds |x| of the input tensor, device memory
    const float* wscale;      // [G][Mpad] inverse row scales written by bm_pack_weights_h2
    BmAmaxDst y_amax;         // where max |y_out| goes (bm_publish_amax): per-workgroup partials
};

// Power-of-two scale s with amax * s in [2^14, 2^15), and its exact inverse.  amax == 0 / subnormal / inf /
// nan: s = 1 (non-finite operands then propagate through the split as inf / nan like in fp32).
__host__ __device__ __forceinline__ void h2_scale_from_amax(float amax, float& s, float& inv) {
    unsigned bits;
    memcpy(&bits, &amax, 4);
    const unsigned e = (bits >> 23) & 0xffu;
    int se = 127;
    if (e != 0u && e != 255u) {
        se = 268 - (int)e;              // 127 + 14 - (e - 127)
        se = se > 253 ? 253 : (se < 1 ? 1 : se);
    }
    const unsigned sb = (unsigned)se << 23, ib = (unsigned)(254 - se) << 23;
    memcpy(&s, &sb, 4);
    memcpy(&inv, &ib, 4);
}

// 8 fp32 values (already scaled) -> f16 planes hi, lo
__device__ __forceinline__ void split8h(const float* f, float s, u32x4& hi, u32x4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float xs = f[i] * s;
        const _Float16 a = (_Float16)xs;
        h[i] = a;
        l[i] = (_Float16)(xs - (float)a);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// two fp32 values -> scaled f16 pairs: hi = f16(x * s), lo = f16(x * s - hi) (the product is exact, s is a power
// of two; the difference is exact in fp32), written straight into the halves of the packed results: 4 VALU
__device__ __forceinline__ void ch_split_pair(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(x0), "v"(x1), "v"(s));
}

template <int... I, class F>
__device__ __forceinline__ void h2_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void h2_static_for(F&& f) {
    h2_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ float ch_ld32(i32x4h rs, int voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}

// Epilogue helpers: every accumulator block is addressed with compile-time indices (template recursion), so
// the MW x 3 accumulators never leave the register file.
template <int MW>
struct H2Simple {
    float* yb;
    const float* rb;
    int rowu, li, T;
    int row0, M;              // first row of this lane (rowu + 4 h), rows of the layer
    float bia[MW][16];
};

template <int MW, int NT, int MT, bool FULL>
__device__ __forceinline__ float h2_simple_col(const H2Simple<MW>& e, f32x16 (&acc)[MW][3], float amx) {
    if constexpr (MT < MW) {
        // row of element r of this lane: e.row0 + MT * 32 + (r & 3) + 8 * (r >> 2); FULL: the tile lies inside M
        float rv[16];
        if (e.rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = MT * 32 + (r & 3) + 8 * (r >> 2);
                rv[r] = (FULL || e.row0 + dr < e.M) ? e.rb[(long)(e.rowu + dr) * e.T + e.li + NT * 32] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = MT * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc[MT][NT][r] + e.bia[MT][r];
            if (e.rb) v += rv[r];
            if (FULL || e.row0 + dr < e.M) {
                amx = fmaxf(amx, fabsf(v));
                e.yb[(long)(e.rowu + dr) * e.T + e.li + NT * 32] = v;
            }
        }
        return h2_simple_col<MW, NT, MT + 1, FULL>(e, acc, amx);
    }
    return amx;
}

template <int MW, bool FULL>
__device__ __forceinline__ float h2_simple_tile(const H2Simple<MW>& e, f32x16 (&acc)[MW][3], int col0, int T) {
    float amx = 0.f;
    if (col0 < T) amx = h2_simple_col<MW, 0, 0, FULL>(e, acc, amx);
    if (col0 + 32 < T) amx = h2_simple_col<MW, 1, 0, FULL>(e, acc, amx);
    if (col0 + 64 < T) amx = h2_simple_col<MW, 2, 0, FULL>(e, acc, amx);
    return amx;
}

template <int MW, int I>
__device__ __forceinline__ float h2_general_blocks(const ConvNNArgs& a, f32x16 (&acc)[MW][3], const float* ep, int b,
                                                   int m0, int rl_base, int col_base, float amx) {
    if constexpr (I < MW * 3) {
        constexpr int MT = I / 3, NT = I % 3;
        float v[16];
        float m = 0.f;
        conv_ep_store_block(a, acc[MT][NT], ep, 64 * MW, b, m0, rl_base + MT * 32, col_base + NT * 32, v, &m);
        return h2_general_blocks<MW, I + 1>(a, acc, ep, b, m0, rl_base, col_base, fmaxf(amx, m));
    }
    return amx;
}

// BatchNorm statistics of the tile (training-mode conv + BN layers, bm/models/common.py:119): per output row, sum and
// sum of squares of y_pre = acc + bias over this wavefront's 96 columns, written as partial (tile, wavefront
// column) of the channel-major `stats` [M][B * ntiles_n * 2][2] -- bm_bn_finalize_cm adds a channel's partials, one
// contiguous run, in double.  Saves the
// channel_stats pass over the 118 MB output.  Per row block: the lane's 3 column blocks are summed in registers,
// then a halving butterfly over the 32 lanes of a half-wavefront (16 + 8 + 4 + 2 + 1 + 1 exchanges for 2 x 16
// values instead of 5 x 32) leaves row q = 8 b4 + 4 b3 + 2 b2 + b1 (bk = bit k of the lane) in each lane.
template <int MW>
__device__ __forceinline__ void h2_tile_stats(const ConvNNArgs& a, f32x16 (&acc)[MW][3], const float* epl /* lane's bias rows */,
                                              float* stats_tile /* + (tile * 2 + wn) * 2 */, long row_stride /* floats between channels */,
                                              int row0 /* first row of the lane's blocks */, int col0, int lane) {
    const bool c0ok = col0 < a.T, c1ok = col0 + 32 < a.T, c2ok = col0 + 64 < a.T;
    // wave-uniform: every column of the wavefront's 96 lies inside T (3 of the 4 wavefront tiles of a T = 360 segment)
    const bool all_in = __builtin_amdgcn_readfirstlane(col0 - (lane & 31)) + 95 < a.T;
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    h2_static_for<MW>([&](auto mc) __attribute__((always_inline)) {
        constexpr int mt = decltype(mc)::value;
        float s1[16], s2[16];
        if (all_in) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bias = epl[mt * 32 + (r & 3) + 8 * (r >> 2)];
                const float v0 = acc[mt][0][r] + bias, v1 = acc[mt][1][r] + bias, v2 = acc[mt][2][r] + bias;
                s1[r] = (v0 + v1) + v2;
                s2[r] = fmaf(v0, v0, fmaf(v1, v1, v2 * v2));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bias = epl[mt * 32 + (r & 3) + 8 * (r >> 2)];
                const float v0 = c0ok ? acc[mt][0][r] + bias : 0.f;
                const float v1 = c1ok ? acc[mt][1][r] + bias : 0.f;
                const float v2 = c2ok ? acc[mt][2][r] + bias : 0.f;
                s1[r] = (v0 + v1) + v2;
                s2[r] = fmaf(v0, v0, fmaf(v1, v1, v2 * v2));
            }
        }
#define H2_FOLD(N_, BIT_, MASK_)                                                                  \
        _Pragma("unroll") for (int i = 0; i < N_; ++i) {                                          \
            const float k1 = BIT_ ? s1[i + N_] : s1[i], g1 = BIT_ ? s1[i] : s1[i + N_];            \
            const float k2 = BIT_ ? s2[i + N_] : s2[i], g2 = BIT_ ? s2[i] : s2[i + N_];            \
            s1[i] = k1 + __shfl_xor(g1, MASK_);                                                   \
            s2[i] = k2 + __shfl_xor(g2, MASK_);                                                   \
        }
        H2_FOLD(8, b4, 16)
        H2_FOLD(4, b3, 8)
        H2_FOLD(2, b2, 4)
        H2_FOLD(1, b1, 2)
#undef H2_FOLD
        s1[0] += __shfl_xor(s1[0], 1);
        s2[0] += __shfl_xor(s2[0], 1);
        // element index q of the lane's 16 -> row (q & 3) + 8 (q >> 2) of the block (C/D layout)
        const int q = (b4 ? 8 : 0) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
        const int row = row0 + mt * 32 + (q & 3) + 8 * (q >> 2);
        if (!(lane & 1) && row < a.M)
            *reinterpret_cast<float2*>(stats_tile + (long)row * row_stride) = float2{s1[0], s2[0]};
    });
}

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
// cycle trace of the stage pipeline (diagnostic builds only, scripts/build_trace_lib.sh): per workgroup and
// wavefront, [tap j][segment] cycles summed over the stages, [3][7] = stage count
__device__ unsigned ch_trace_buf[64 * 4 * 24];
#define CH_T(I_) { __builtin_amdgcn_sched_barrier(0); tr[I_] = (unsigned)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int bm_debug_trace_read_conv(unsigned* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_trace_buf), sizeof(unsigned) * 64 * 4 * 24);
}
#else
#define CH_T(I_)
#endif

template <int KS, int MW>
__global__ __launch_bounds__(256, 1) void conv_nn_h2w_kernel(ConvH2Args args) {
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

    // epilogue: inverse scales (exact powers of two), bias, optional pre-activation store, per-channel affine,
    // activation, residual.  One literal-indexed expansion per accumulator block keeps the accumulators in
    // registers.  ep[3 HBM .. 4 HBM) = per-row accumulator factor = inverse weight-row scale x inverse x scale.
    float* ep = smem;                // the operand buffers are free after the last barrier of the main loop
    conv_ep_stage_params(a, ep, HBM, m0, tid, 256, b);
    {
        const float* ws = args.wscale + (long)g * a.Mpad + m0;
        for (int i = tid; i < HBM; i += 256) ep[3 * HBM + i] = ws[i] * sx_inv;
    }
    __syncthreads();
    {
        const float* fl = ep + 3 * HBM + wm * (MW * 32) + 4 * h;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = fl[mt * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                for (int nt = 0; nt < NW; ++nt) acc[mt][nt][r] *= f;
            }
    }
    // Common case (every wide launch of the training step): one output tensor, no affine / activation.  Row
    // addresses are wave-uniform bases + one per-lane offset + an immediate per column block, the row biases are
    // read once: ~3 VALU issue slots per stored element instead of ~10.  Tiles that straddle M (270-channel
    // layers in 320-row tiles) take the same path with a row predicate.
    if (a.stats)            // training-mode BatchNorm layers: the host allows it only with y_pre alone (simple path)
        h2_tile_stats<MW>(a, acc, ep + wm * (MW * 32) + 4 * h,
                          a.stats + (long)((b * a.ntiles_n + ntile) * 2 + wn) * 2, (long)a.B * a.ntiles_n * 4,
                          m0 + wm * (MW * 32) + 4 * h, n0 + wn * (NW * 32) + nl, lane);
    const bool simple = ((a.y_pre != nullptr) != (a.y_out != nullptr)) && !a.ep_scale && a.act == BM_ACT_NONE;
    if (simple) {
        H2Simple<MW> e;
        e.yb = (a.y_pre ? a.y_pre : a.y_out) + (long)b * a.y_bstride;
        e.rb = (a.y_out && a.res) ? a.res + (long)b * a.res_bstride : nullptr;
        e.rowu = m0 + wm * (MW * 32);                       // wave-uniform first row
        e.li = 4 * h * a.T + n0 + wn * (NW * 32) + nl;      // per-lane element offset inside a row block
        e.T = a.T;
        e.row0 = e.rowu + 4 * h;
        e.M = a.M;
        const float* epl = ep + wm * (MW * 32) + 4 * h;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) e.bia[mt][r] = epl[mt * 32 + (r & 3) + 8 * (r >> 2)];
        const int col0 = n0 + wn * (NW * 32) + nl;
        const float amx = m0 + HBM <= a.M ? h2_simple_tile<MW, true>(e, acc, col0, a.T)
                                          : h2_simple_tile<MW, false>(e, acc, col0, a.T);
        if (a.y_out) bm_publish_amax(amx, args.y_amax, smem + 4 * HBM);
    } else {
        const float amx = h2_general_blocks<MW, 0>(a, acc, ep, b, m0, wm * (MW * 32) + 4 * h,
                                                   n0 + wn * (NW * 32) + nl, 0.f);
        bm_publish_amax(amx, args.y_amax, smem + 4 * HBM);
    }
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

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// rows per wavefront-row block: the workgroup tile is 64 MW rows; MW in {5, 4, 2} minimising the padded rows
// (ties -> the larger tile)
extern "C" int bm_conv_h2_mw_for(int M) {
    int best = 5;
    long best_pad = -1;
    const int cand[3] = {5, 4, 2};
    for (int i = 0; i < 3; ++i) {
        const long bm = 64L * cand[i];
        const long pad = (long)cdiv(M, bm) * bm;
        if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = cand[i]; }
    }
    return best;
}
extern "C" int bm_conv_h2_mpad(int M) {
    const int bm = 64 * bm_conv_h2_mw_for(M);
    return cdiv(M, bm) * bm;
}

// 1 when the wide f16x2 kernel covers this conv shape (otherwise the caller packs for / launches the 3 x bf16
// kernels of conv_nn_x3.hip, which are fp32-accurate as well)
extern "C" int bm_conv_h2_covers(int Cin, int M, int T, int KS, int dil) {
    if (KS != 1 && KS != 3) return 0;
    if ((KS >> 1) * dil > 16 || T <= 128) return 0;
    if ((long)Cin * T * 4 >= 0x40000000L) return 0;
    if ((long)Cin * KS > 3840) return 0;              // PACK_MAX_NK: a weight row must fit the packing kernel's LDS tile
    // padded rows are wasted MFMA work: leave tiny layers (tests, F = 16 heads) to the narrow kernels
    return (long)bm_conv_h2_mpad(M) * 2 <= (long)M * 3 || M >= 96;
}

// rows of the `stats` buffer of bm_conv1d_nn_h2: one per (segment, column tile, wavefront column)
extern "C" int bm_conv_h2_stats_tiles(int B, int T) { return B * cdiv(T, HBN) * 2; }

// bytes of the packed buffer: f16 planes [G][chunk32][KS][2][4][Mpad][8] + fp32 inverse row scales [G][Mpad]
extern "C" long bm_packed_weight_bytes_h2(int G, int M, int Cin, int KS) {
    const long mpad = bm_conv_h2_mpad(M);
    return (long)G * cdiv(Cin, 32) * KS * 2 * 4 * mpad * 8 * 2 + (long)G * mpad * 4;
}

// One workgroup per (group, padded row): row maximum -> power-of-two scale -> the row's 16-byte slots of both
// planes (zeros for padded rows / channels) and its inverse scale.
struct PackH2Job {
    const float* src;
    unsigned short* dst;
    float* wscale;
    const float* alpha;
    long sg, sm, sc, sj;
    int M, Cin, KS, flip, Mpad, nchunk;
    int block0, nblocks;      // workgroups [block0, block0 + nblocks) of a batched launch belong to this job
};

// PACK_ROWS consecutive rows per workgroup of 256 threads (one wavefront per row), staged through LDS so that both
// source layouts are read along their contiguous direction: the forward layout a whole row at a time, the
// transposed (data-gradient) layout PACK_ROWS x KS contiguous floats per channel -- read element by element it
// touched a different cache line with every value and the 36 MB of parameters took 175 us to pack.
#define PACK_ROWS 4
#define PACK_MAX_NK 3840      // Cin * KS values of one row that the LDS tile holds (61 KB)

__device__ __forceinline__ void pack_h2_rows(const PackH2Job& jb, int block, float* tile, float* rs) {
    const float* __restrict__ src = jb.src;
    const int M = jb.M, Cin = jb.Cin, KS = jb.KS, Mpad = jb.Mpad, nchunk = jb.nchunk, flip = jb.flip;
    const long sg = jb.sg, sm = jb.sm, sc = jb.sc, sj = jb.sj;
    const int rows_per_group = Mpad / PACK_ROWS;
    const int g = block / rows_per_group, m0 = (block - g * rows_per_group) * PACK_ROWS;
    const float alpha = jb.alpha ? *jb.alpha : 1.f;
    const int nk = Cin * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1. tile[r][c * KS + j] = alpha * w[m0 + r][c][j]; the faster of (row, channel) in memory runs inside
    const float* base = src + g * sg + (long)m0 * sm;
    const bool rows_inside = sm < sc;
    for (int idx = tid; idx < PACK_ROWS * nk; idx += blockDim.x) {
        int r, c, j;
        if (rows_inside) {                  // idx = (c * PACK_ROWS + r) * KS + j
            j = idx % KS;
            const int t = idx / KS;
            r = t % PACK_ROWS;
            c = t / PACK_ROWS;
        } else {                            // idx = (r * Cin + c) * KS + j
            j = idx % KS;
            const int t = idx / KS;
            c = t % Cin;
            r = t / Cin;
        }
        tile[r * nk + c * KS + j] = (m0 + r < M) ? alpha * base[r * sm + c * sc + j * sj] : 0.f;
    }
    __syncthreads();
    // 2. row maxima -> scales (wavefront w owns row w)
    {
        float mx = 0.f;
        for (int i = lane; i < nk; i += 64) mx = fmaxf(mx, fabsf(tile[wave * nk + i]));
        mx = bm_wave_max(mx);
        float s, inv;
        h2_scale_from_amax(mx, s, inv);
        if (lane == 0) {
            rs[wave] = s;
            jb.wscale[(long)g * Mpad + m0 + wave] = inv;
        }
    }
    __syncthreads();
    // 3. the rows' 16-byte slots of both planes: slot q = (chunk, tap, 8-channel group); rows run fastest so that
    //    neighbouring threads write neighbouring 16-byte slots
    const int nslots = nchunk * KS * 4;
    const long plane_stride = (long)4 * Mpad * 8;           // f16 elements of one plane of one (chunk, tap)
    for (int t = tid; t < PACK_ROWS * nslots; t += blockDim.x) {
        const int r = t % PACK_ROWS, q = t / PACK_ROWS;
        const int kg = q & 3;
        const int j = (q >> 2) % KS;
        const int chunk = (q >> 2) / KS;
        const int jj = flip ? KS - 1 - j : j;
        const float sr = rs[r];
        alignas(16) unsigned short hi[8];
        alignas(16) unsigned short lo[8];
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const int c = chunk * 32 + kg * 8 + e8;
            const float v = c < Cin ? tile[r * nk + c * KS + jj] * sr : 0.f;
            const _Float16 a = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)a);
            hi[e8] = __builtin_bit_cast(unsigned short, a);
            lo[e8] = __builtin_bit_cast(unsigned short, l);
        }
        const long stage = ((long)g * nchunk + chunk) * KS + j;
        unsigned short* out = jb.dst + stage * 2 * plane_stride + ((long)kg * Mpad + m0 + r) * 8;
        *reinterpret_cast<uint4*>(out) = *reinterpret_cast<const uint4*>(hi);
        *reinterpret_cast<uint4*>(out + plane_stride) = *reinterpret_cast<const uint4*>(lo);
    }
}

__global__ __launch_bounds__(256) void pack_weights_h2_kernel(PackH2Job jb) {
    extern __shared__ float pack_smem[];
    pack_h2_rows(jb, blockIdx.x, pack_smem + PACK_ROWS, pack_smem);
}

// every weight tensor of a model in ONE launch: `jobs` (device memory) sorted by block0
__global__ __launch_bounds__(256) void pack_weights_h2_batch_kernel(const PackH2Job* __restrict__ jobs, int njobs) {
    extern __shared__ float pack_smem[];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                   // last job with block0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackH2Job jb = jobs[lo];
    pack_h2_rows(jb, blockIdx.x - jb.block0, pack_smem + PACK_ROWS, pack_smem);
}

static int pack_h2_fill(PackH2Job& jb, const float* src, void* dst, int G, int M, int Cin, int KS, long sg, long sm,
                        long sc, long sj, int flip, const float* alpha_ptr) {
    BM_REQUIRE(src && dst, "pack_weights_h2: null pointer");
    BM_REQUIRE(G > 0 && M > 0 && Cin > 0 && KS > 0, "pack_weights_h2: bad dims");
    jb.Mpad = bm_conv_h2_mpad(M);
    jb.nchunk = cdiv(Cin, 32);
    const long f16_elems = (long)G * jb.nchunk * KS * 2 * 4 * jb.Mpad * 8;
    jb.src = src; jb.dst = (unsigned short*)dst;
    jb.wscale = reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + f16_elems * 2);
    jb.alpha = alpha_ptr;
    jb.sg = sg; jb.sm = sm; jb.sc = sc; jb.sj = sj;
    jb.M = M; jb.Cin = Cin; jb.KS = KS; jb.flip = flip;
    BM_REQUIRE(Cin * KS <= PACK_MAX_NK, "pack_weights_h2: Cin * KS = %d exceeds %d", Cin * KS, PACK_MAX_NK);
    jb.block0 = 0; jb.nblocks = G * jb.Mpad / PACK_ROWS;
    return BM_OK;
}

extern "C" int bm_pack_weights_h2(const float* src, void* dst, int G, int M, int Cin, int KS, long sg, long sm,
                                  long sc, long sj, int flip, const float* alpha_ptr, void* stream) {
    PackH2Job jb;
    if (int rc = pack_h2_fill(jb, src, dst, G, M, Cin, KS, sg, sm, sc, sj, flip, alpha_ptr)) return rc;
    hipLaunchKernelGGL(pack_weights_h2_kernel, dim3((unsigned)jb.nblocks), dim3(256),
                       (size_t)(PACK_ROWS + PACK_ROWS * Cin * KS) * sizeof(float), (hipStream_t)stream, jb);
    return bm_check_launch("pack_weights_h2");
}

// Batched packing.  The host builds a table of jobs (bm_pack_h2_job_bytes() bytes each, filled by
// bm_pack_h2_job_fill, which returns the job's workgroup count or a negative error), copies it to the device
// once, and re-packs every weight tensor of the model with one launch per optimizer step.
extern "C" int bm_pack_h2_job_bytes() { return (int)sizeof(PackH2Job); }

extern "C" int bm_pack_h2_job_fill(void* job, const float* src, void* dst, int G, int M, int Cin, int KS, long sg,
                                   long sm, long sc, long sj, int flip, const float* alpha_ptr, int block0) {
    if (!job) return -1;
    PackH2Job jb;
    if (pack_h2_fill(jb, src, dst, G, M, Cin, KS, sg, sm, sc, sj, flip, alpha_ptr)) return -1;
    jb.block0 = block0;
    memcpy(job, &jb, sizeof(jb));
    return jb.nblocks;
}

extern "C" int bm_pack_weights_h2_batch(const void* jobs_dev, int njobs, int total_blocks, int max_nk, void* stream) {
    BM_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "pack_weights_h2_batch: bad arguments");
    BM_REQUIRE(max_nk > 0 && max_nk <= PACK_MAX_NK, "pack_weights_h2_batch: max_nk = %d (largest Cin * KS of the jobs)", max_nk);
    hipLaunchKernelGGL(pack_weights_h2_batch_kernel, dim3((unsigned)total_blocks), dim3(256),
                       (size_t)(PACK_ROWS + PACK_ROWS * max_nk) * sizeof(float), (hipStream_t)stream,
                       (const PackH2Job*)jobs_dev, njobs);
    return bm_check_launch("pack_weights_h2_batch");
}

// max |x[i]| as <= 256 per-workgroup partial maxima (ws) folded by bm_amax_finalize into the slot `out`
// (BM_AMAX_SHARDS floats whose maximum is the answer).  A non-finite input yields a non-finite or NaN-free
// maximum of the finite part; the non-finite values themselves propagate through the consumer's split.
__global__ __launch_bounds__(1024) void amax_kernel(const float* __restrict__ x, long n, float* __restrict__ ws,
                                                    int* __restrict__ nonfinite) {
    __shared__ float sh[16];
    float mx = 0.f;
    unsigned top = 0u;                                  // largest |x| bit pattern: >= 0x7f800000 <=> inf / nan seen
    // a tensor that is only 4-byte aligned (a batch slice such as meg[1:] with C * T odd): `head` scalar elements
    // up to the first 16-byte boundary, the vector body from there, the scalar tail behind it
    long head = (long)((16u - (unsigned)((uintptr_t)x & 15u)) & 15u) >> 2;
    if (head > n) head = n;
    const long n4 = (n - head) >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + head);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
        top = max(max(top, __float_as_uint(v.x) & 0x7fffffffu),
                  max(__float_as_uint(v.y) & 0x7fffffffu, max(__float_as_uint(v.z) & 0x7fffffffu,
                                                               __float_as_uint(v.w) & 0x7fffffffu)));
    }
    if (blockIdx.x == 0) {
        const long tail0 = head + (n4 << 2);
        for (long i = threadIdx.x; i < head + (n - tail0); i += blockDim.x) {
            const long k = i < head ? i : tail0 + (i - head);
            mx = fmaxf(mx, fabsf(x[k]));
            top = max(top, __float_as_uint(x[k]) & 0x7fffffffu);
        }
    }
    if (nonfinite && top >= 0x7f800000u) atomicOr(nonfinite, 1);     // rare: at most one atomic per thread
    bm_publish_amax(mx, ws, sh);
}

// `nonfinite_flag` (nullable, device int): set to 1 if x holds an inf or a nan -- the reference's
// `torch.isfinite(x).all()` asserts (bm/solver.py:258-260) ride on the pass that the f16x2 scale needs anyway.
extern "C" int bm_amax_checked(const float* x, long n, float* out, float* ws, int* nonfinite_flag, void* stream) {
    BM_REQUIRE(x && out && ws && n >= 0, "amax: bad arguments");
    BM_REQUIRE(((uintptr_t)x & 3) == 0, "amax: x must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    long blocks = (n / 4 + 1023) / 1024;
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, x, n, ws, nonfinite_flag);
    if (int rc = bm_check_launch("amax")) return rc;
    return bm_amax_finalize(ws, (int)blocks, out, s);
}

extern "C" int bm_amax(const float* x, long n, float* out, float* ws, void* stream) {
    return bm_amax_checked(x, n, out, ws, nullptr, stream);
}

template <int KS, int MW>
static int launch_conv_nn_h2w(ConvH2Args args, float* y_amax_out, hipStream_t stream) {
    constexpr int HBM = 64 * MW;
    size_t lds = (size_t)(3 * 2 * 2 * HBM + 2 * HXSLAB) * 16;
    const size_t lds_ep = (size_t)(4 * HBM + 4) * sizeof(float);
    if (lds < lds_ep) lds = lds_ep;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_h2w_kernel<KS, MW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "conv_nn_h2w: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    args.c.ntiles_n = cdiv(args.c.T, HBN);
    args.c.ntiles_m = args.c.Mpad / HBM;
    const long nblocks = (long)args.c.B * args.c.ntiles_n * args.c.ntiles_m;
    if (nblocks <= 0) return BM_OK;
    // the workspace bounds the grid
    const bool publish = y_amax_out && args.y_amax.ws && nblocks <= BM_AMAX_WS;
    float* ws = args.y_amax.ws;
    if (!publish) args.y_amax = BmAmaxDst{nullptr};
    hipLaunchKernelGGL((conv_nn_h2w_kernel<KS, MW>), dim3((unsigned)nblocks), dim3(256), lds, stream, args);
    if (int rc = bm_check_launch("conv_nn_h2w")) return rc;
    if (publish) return bm_amax_done(args.y_amax, (int)nblocks, y_amax_out, stream);
    if (y_amax_out && ws)           // grid larger than the workspace: a plain pass over the output instead
        return bm_amax(args.c.y_out, (long)args.c.B * args.c.M * args.c.T, y_amax_out, ws, stream);
    return BM_OK;
}

// Same contract as bm_conv1d_nn plus `x_amax` (device pointer to max |x|, e.g. from bm_amax; any upper bound
// within a factor 2 of the fp16 range works) and `y_amax_out` (nullable amax slot: receives max |y_out| for the
// contraction that consumes the output; `amax_ws` = BM_AMAX_WS floats of scratch shared by all producers of the stream); weights packed by bm_pack_weights_h2; only shapes for which
// bm_conv_h2_covers() is 1.
extern "C" int bm_conv1d_nn_h2(const float* x, long x_bstride, const float* x_amax, const void* wpacked,
                               const int* widx, const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift,
                               const float* res, long res_bstride, float* y_pre, float* y_out, long y_bstride,
                               float* stats, int B, int Cin, int M, int T, int KS, int dil, int act, float leak,
                               int G, float* y_amax_out, float* amax_ws, void* stream) {
    BM_REQUIRE(x && wpacked && x_amax, "conv1d_nn_h2: null x / w / x_amax");
    BM_REQUIRE(y_pre || y_out, "conv1d_nn_h2: no output");
    BM_REQUIRE(!stats || (y_pre && !y_out && !ep_scale && act == BM_ACT_NONE && !res),
               "conv1d_nn_h2: per-tile statistics come with y_pre alone (the training-mode BatchNorm layers)");
    BM_REQUIRE(B >= 0 && Cin > 0 && M > 0 && T > 0 && dil >= 1 && G >= 1, "conv1d_nn_h2: bad dims");
    BM_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), "conv1d_nn_h2: scale/shift must come together");
    BM_REQUIRE(bm_conv_h2_covers(Cin, M, T, KS, dil), "conv1d_nn_h2: shape not covered (Cin=%d M=%d T=%d KS=%d dil=%d)",
               Cin, M, T, KS, dil);
    ConvH2Args args;
    ConvNNArgs& a = args.c;
    a.x = x; a.x_bstride = x_bstride; a.wp = (const float*)wpacked; a.widx = widx; a.bias = bias; a.bias_gstride = bias_gstride;
    a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.res = res; a.res_bstride = res_bstride;
    a.y_pre = y_pre; a.y_out = y_out; a.y_bstride = y_bstride; a.stats = stats;
    a.B = B; a.Cin = Cin; a.M = M; a.T = T; a.KS = KS; a.dil = dil; a.act = act; a.leak = leak;
    const int mw = bm_conv_h2_mw_for(M);
    a.Mpad = bm_conv_h2_mpad(M);
    a.nchunk = cdiv(Cin, 32);
    const long f16_elems = (long)G * a.nchunk * KS * 2 * 4 * a.Mpad * 8;
    args.x_amax = x_amax;
    BM_REQUIRE(!y_amax_out || amax_ws, "conv1d_nn_h2: y_amax_out needs the amax workspace");
    if (!y_out) y_amax_out = nullptr;
    args.y_amax = bm_amax_dst(y_amax_out, amax_ws);
    args.wscale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpacked) + f16_elems * 2);
    hipStream_t s = (hipStream_t)stream;
#define H2_DISPATCH(MW_)                                                                 \
    return KS == 1 ? launch_conv_nn_h2w<1, MW_>(args, y_amax_out, s) : launch_conv_nn_h2w<3, MW_>(args, y_amax_out, s);
    switch (mw) {
        case 5: H2_DISPATCH(5)
        case 4: H2_DISPATCH(4)
        default: H2_DISPATCH(2)
    }
#undef H2_DISPATCH
}
