// Shared argument block and output epilogue of the implicit-GEMM conv kernels (fp32 and bf16 MFMA).
#pragma once
#include "bm_common.h"

struct ConvNNArgs {
    const float* x;       long x_bstride;      // [B][Cin][T]
    const float* wp;                           // packed [G][nchunk][KS][BKC][Mpad]
    const int* widx;                           // [B] weight-group per segment, or null (group 0)
    const float* bias;                         // [M] or null
    long bias_gstride;                         // 0, or the stride between the bias vectors of weight groups (widx)
    const float* ep_scale; const float* ep_shift;   // [M] per-channel affine applied before act, or null
    const float* res;     long res_bstride;    // residual added after act, or null
    float* y_pre;                              // pre-activation output (after bias), or null
    float* y_out;                              // post-epilogue output, or null
    long y_bstride;
    float* stats;                              // [B*NTILES][M][2] per-tile (sum, sumsq) of y_pre, or null
    int B, Cin, M, T, KS, dil, Mpad, nchunk, act;
    float leak;
    int ntiles_n, ntiles_m;
};

// Per-row epilogue parameters of a tile (bias, affine scale, affine shift) staged once into LDS:
// ep[0..BM) = bias, ep[BM..2BM) = scale, ep[2BM..3BM) = shift.  Reading them from global memory per
// element serialises hundreds of dependent L2 round trips per wavefront at the end of every tile.
__device__ __forceinline__ void conv_ep_stage_params(const ConvNNArgs& a, float* ep, int BM, int m0, int tid,
                                                     int nthreads, int b) {
    // per-group bias (the composed front end: bias = W_subject^T b_initial differs per subject)
    const float* bias = a.bias;
    if (bias && a.bias_gstride && a.widx) bias += (long)a.widx[b] * a.bias_gstride;
    for (int i = tid; i < BM; i += nthreads) {
        const int row = m0 + i;
        const bool ok = row < a.M;
        ep[i] = (bias && ok) ? bias[row] : 0.f;
        ep[BM + i] = (a.ep_scale && ok) ? a.ep_scale[row] : 1.f;
        ep[2 * BM + i] = (a.ep_shift && ok) ? a.ep_shift[row] : 0.f;
    }
}

// One 32x32 accumulator block (C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)):
// bias, optional pre-activation store, per-channel affine, activation, residual.  `rl0` = tile-local row of
// element r = 0 of this lane (includes the 4 * (lane >> 5) term).  The column predicate is applied once
// around the block, the row predicate only in tiles that straddle M (FULL = false), output pointers are
// wave-uniform bases + one 32-bit lane offset, and the residual values of the block are fetched as one
// batch of independent loads before anything is stored.
template <bool FULL>
__device__ __forceinline__ void conv_ep_block_stores(const ConvNNArgs& a, const float* ep, int BM, int b,
                                                     int m0, int rl0, int col, const float (&v)[16],
                                                     float* amx = nullptr) {
    const int row0 = m0 + rl0;
    const int base = row0 * a.T + col;
    float rv[16];
    if (a.res) {
        const float* rb = a.res + (long)b * a.res_bstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            rv[r] = (FULL || row0 + dr < a.M) ? rb[base + dr * a.T] : 0.f;
        }
    }
    if (a.y_pre) {
        float* yp = a.y_pre + (long)b * a.y_bstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (FULL || row0 + dr < a.M) yp[base + dr * a.T] = v[r];
        }
    }
    if (a.y_out) {
        float* yo = a.y_out + (long)b * a.y_bstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const int rl = rl0 + dr;
            float z = v[r] * ep[BM + rl] + ep[2 * BM + rl];
            z = bm_act(z, a.act, a.leak);
            if (a.res) z += rv[r];
            if (FULL || row0 + dr < a.M) {
                yo[base + dr * a.T] = z;
                if (amx) *amx = fmaxf(*amx, fabsf(z));      // max |y_out| for the f16x2 scale of the consumer
            }
        }
    }
}

__device__ __forceinline__ void conv_ep_store_block(const ConvNNArgs& a, const f32x16& c, const float* ep,
                                                    int BM, int b, int m0, int rl0, int col, float (&v)[16],
                                                    float* amx = nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = c[r] + ep[rl0 + (r & 3) + 8 * (r >> 2)];
    if (col < a.T) {
        if (m0 + BM <= a.M) conv_ep_block_stores<true>(a, ep, BM, b, m0, rl0, col, v, amx);
        else conv_ep_block_stores<false>(a, ep, BM, b, m0, rl0, col, v, amx);
    }
}

// Epilogue for a [32*MT] x [128] tile held as MT 32x32 MFMA accumulators per wavefront (4 wavefronts,
// wavefront w owns columns [32w, 32w+32)): bias, optional pre-activation store, optional per-tile
// BatchNorm partial statistics, optional per-channel affine, activation, residual.
// `tid` is the thread index inside the 4-wavefront group that owns the tile; `smem` is the workgroup's LDS,
// free by now: [4 waves][BM][2] statistics scratch, then the 3 * BM staged row parameters.
template <int MT>
__device__ __forceinline__ void conv_tile_epilogue(const ConvNNArgs& a, f32x16 (&acc)[MT], float* smem,
                                                   int b, int ntile, int m0, int n0, int tid) {
    constexpr int BM = 32 * MT;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nl = lane & 31;
    const int h = lane >> 5;
    const int col = n0 + wave * 32 + nl;
    const bool col_ok = col < a.T;
    float* red = smem;
    float* ep = smem + 8 * BM;
    __syncthreads();                 // every wavefront is done with the operand tiles
    conv_ep_stage_params(a, ep, BM, m0, tid, 256, b);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float v[16];
        conv_ep_store_block(a, acc[mt], ep, BM, b, m0, mt * 32 + 4 * h, col, v);
        if (a.stats) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float s = (col_ok && m0 + rl < a.M) ? v[r] : 0.f;
                float s2 = s * s;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    s += __shfl_xor(s, o);
                    s2 += __shfl_xor(s2, o);
                }
                if (nl == 0) {
                    red[(wave * BM + rl) * 2 + 0] = s;
                    red[(wave * BM + rl) * 2 + 1] = s2;
                }
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        for (int rl = tid; rl < BM; rl += 256) {
            const int row = m0 + rl;
            if (row < a.M) {
                float s = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s += red[(w * BM + rl) * 2 + 0];
                    s2 += red[(w * BM + rl) * 2 + 1];
                }
                float* dst = a.stats + ((long)(b * a.ntiles_n + ntile) * a.M + row) * 2;
                dst[0] = s;
                dst[1] = s2;
            }
        }
    }
}
