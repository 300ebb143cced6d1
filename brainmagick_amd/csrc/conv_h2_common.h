// Shared by the two wide f16x2 conv translation units (conv_nn_h2w.hip: the production main loop, conv_nn_h2d.hip: the
// round-2..5 main loop kept for A/B runs): argument block, scale rule, operand split, tile epilogue.
#pragma once
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

// Residual tiles (the data-gradient convs of the residual layers: y = conv + bias + res).  Fetched block by block in
// front of each block's stores -- 16 loads, wait, 16 stores -- a load's wait also waits for the OLDER stores (one in-order
// counter), so the tile paid fifteen load + store round trips to HBM: 61-67 K cycles per tile against 28 K without a
// residual (cycle trace, profiles/r6_trace_conv.txt).  Two phases instead: (1) the residual values are added INTO the
// finished accumulators, loads only, three blocks (48 dword loads per lane) in flight; (2) the stores.  The order of
// the additions is the old one, (acc x scale + bias) + res.  (`y` may be the residual tensor itself: every element is
// read in phase 1 and written in phase 2 by the same thread.)
struct H2Plain {
    float* yb;
    const float* rb;
    int rowu, li, T;
    int row0, M;
};

template <int MW, bool FULL>
__device__ __forceinline__ void h2_residual_add(const H2Plain& e, f32x16 (&acc)[MW][3], int col0, int T) {
    constexpr int NBLK = MW * 3, D = 3;
    float rv[D][16];
    const bool cok[3] = {col0 < T, col0 + 32 < T, col0 + 64 < T};
    h2_static_for<NBLK + D>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i >= D) {                                   // block i - D has landed
            constexpr int b = i - D, nt = b / MW, mt = b % MW;
            if (cok[nt]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] += rv[b % D][r];
            }
        }
        if constexpr (i < NBLK) {                                 // request block i
            constexpr int nt = i / MW, mt = i % MW;
            if (cok[nt]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = mt * 32 + (r & 3) + 8 * (r >> 2);
                    rv[i % D][r] = (FULL || e.row0 + dr < e.M) ? e.rb[(long)(e.rowu + dr) * e.T + e.li + nt * 32] : 0.f;
                }
            }
        }
    });
}
// the stores of finished accumulators (no bias, no residual left to add)
template <int MW, int NT, int MT, bool FULL>
__device__ __forceinline__ float h2_plain_col(const H2Plain& e, f32x16 (&acc)[MW][3], float amx) {
    if constexpr (MT < MW) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = MT * 32 + (r & 3) + 8 * (r >> 2);
            const float v = acc[MT][NT][r];
            if (FULL || e.row0 + dr < e.M) {
                amx = fmaxf(amx, fabsf(v));
                e.yb[(long)(e.rowu + dr) * e.T + e.li + NT * 32] = v;
            }
        }
        return h2_plain_col<MW, NT, MT + 1, FULL>(e, acc, amx);
    }
    return amx;
}
template <int MW, bool FULL>
__device__ __forceinline__ float h2_plain_tile(const H2Plain& e, f32x16 (&acc)[MW][3], int col0, int T) {
    float amx = 0.f;
    if (col0 < T) amx = h2_plain_col<MW, 0, 0, FULL>(e, acc, amx);
    if (col0 + 32 < T) amx = h2_plain_col<MW, 1, 0, FULL>(e, acc, amx);
    if (col0 + 64 < T) amx = h2_plain_col<MW, 2, 0, FULL>(e, acc, amx);
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

// Epilogue of a tile, shared by the two main loops below: inverse scales (exact powers of two), bias, optional
// pre-activation store, per-channel affine, activation, residual, BatchNorm partial sums, max |y|.  One literal-indexed
// expansion per accumulator block keeps the accumulators in registers.  `smem` is the workgroup's LDS, free by now
// (the caller's last barrier): ep[0 .. 3 HBM) = staged row parameters, ep[3 HBM .. 4 HBM) = per-row accumulator
// factor = inverse weight-row scale x inverse x scale.
template <int MW, bool RESK = false>
__device__ __forceinline__ void h2_tile_epilogue(const ConvH2Args& args, f32x16 (&acc)[MW][3], float* smem, int b, int g,
                                                 int m0, int n0, int ntile, float sx_inv, int tid, int lane, int wm,
                                                 int wn, int nl, int h, unsigned* trace_out = nullptr) {
    const ConvNNArgs& a = args.c;
    constexpr int NW = 3;
    constexpr int HBM = 64 * MW;
#ifdef HG_TRACE
    unsigned te[6];
#define H2_TE(I_) { __builtin_amdgcn_sched_barrier(0); te[I_] = (unsigned)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define H2_TE(I_)
#endif
    H2_TE(0)
    float* ep = smem;                // the operand buffers are free after the last barrier of the main loop
    conv_ep_stage_params(a, ep, HBM, m0, tid, 256, b);
    {
        const float* ws = args.wscale + (long)g * a.Mpad + m0;
        for (int i = tid; i < HBM; i += 256) ep[3 * HBM + i] = ws[i] * sx_inv;
    }
    __syncthreads();
    H2_TE(1)
    if constexpr (RESK) {
        // The residual kernel: the host launches it only for y_out = conv + bias + res with nothing else in the epilogue,
        // so this is ALL of it, straight-line: finish the accumulators ((acc x scale) + bias: the fma rounds like the two
        // steps, acc x scale is exact), add the residual values (loads only, three blocks in flight), store.  The same
        // updates behind a run-time `if (res)` made hipcc keep two copies of the accumulators (1.3 KB of scratch per
        // lane, the conv 50 % slower).
        const float* fl = ep + 3 * HBM + wm * (MW * 32) + 4 * h;
        const float* bl = ep + wm * (MW * 32) + 4 * h;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = fl[mt * 32 + (r & 3) + 8 * (r >> 2)];
                const float bb = bl[mt * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                for (int nt = 0; nt < NW; ++nt) acc[mt][nt][r] = fmaf(acc[mt][nt][r], f, bb);
            }
        H2Plain e;
        e.yb = a.y_out + (long)b * a.y_bstride;
        e.rb = a.res + (long)b * a.res_bstride;
        e.rowu = m0 + wm * (MW * 32);                       // wave-uniform first row
        e.li = 4 * h * a.T + n0 + wn * (NW * 32) + nl;      // per-lane element offset inside a row block
        e.T = a.T;
        e.row0 = e.rowu + 4 * h;
        e.M = a.M;
        const int col0 = n0 + wn * (NW * 32) + nl;
        float amx;
        if (m0 + HBM <= a.M) {
            h2_residual_add<MW, true>(e, acc, col0, a.T);
            amx = h2_plain_tile<MW, true>(e, acc, col0, a.T);
        } else {
            h2_residual_add<MW, false>(e, acc, col0, a.T);
            amx = h2_plain_tile<MW, false>(e, acc, col0, a.T);
        }
        bm_publish_amax(amx, args.y_amax, smem + 4 * HBM);
        return;
    }
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
    H2_TE(2)
    // Common case (every wide launch of the training step): one output tensor, no affine / activation.  Row
    // addresses are wave-uniform bases + one per-lane offset + an immediate per column block, the row biases are
    // read once: ~3 VALU issue slots per stored element instead of ~10.  Tiles that straddle M (270-channel
    // layers in 320-row tiles) take the same path with a row predicate.
    if (a.stats)            // training-mode BatchNorm layers: the host allows it only with y_pre alone (simple path)
        h2_tile_stats<MW>(a, acc, ep + wm * (MW * 32) + 4 * h,
                          a.stats + (long)((b * a.ntiles_n + ntile) * 2 + wn) * 2, (long)a.B * a.ntiles_n * 4,
                          m0 + wm * (MW * 32) + 4 * h, n0 + wn * (NW * 32) + nl, lane);
    const bool simple = ((a.y_pre != nullptr) != (a.y_out != nullptr)) && !a.ep_scale && a.act == BM_ACT_NONE;
    H2_TE(3)
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
        H2_TE(4)
        if (a.y_out) bm_publish_amax(amx, args.y_amax, smem + 4 * HBM);
        H2_TE(5)
#ifdef HG_TRACE
        if (trace_out && blockIdx.x < 64 && lane == 0) {       // [0][1..3], [1][1..2]: staging, scales, BatchNorm sums, stores, amax
            unsigned* o = trace_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 3) * 8;
            o[1] = te[1] - te[0]; o[2] = te[2] - te[1]; o[3] = te[3] - te[2]; o[8 + 1] = te[4] - te[3]; o[8 + 2] = te[5] - te[4];
        }
#endif
    } else {
        const float amx = h2_general_blocks<MW, 0>(a, acc, ep, b, m0, wm * (MW * 32) + 4 * h,
                                                   n0 + wn * (NW * 32) + nl, 0.f);
        bm_publish_amax(amx, args.y_amax, smem + 4 * HBM);
    }
}


#ifdef HG_TRACE
// cycle trace of the stage pipeline (diagnostic builds only, scripts/build_trace_lib.sh): per workgroup and
// wavefront, [tap j][segment] cycles summed over the stages, [3][7] = stage count; every translation unit has its own
// buffer `ch_trace_buf` (no relocatable device code)
#define CH_T(I_) { __builtin_amdgcn_sched_barrier(0); tr[I_] = (unsigned)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define CH_T(I_)
#endif

// conv_nn_h2d.hip: launch of the LDS-DMA-staged main loop (BM_CONV_LDSDMA=1)
int bm_launch_conv_nn_h2d(const ConvH2Args& args, int KS, int mw, size_t lds_bytes, unsigned nblocks, hipStream_t stream);
