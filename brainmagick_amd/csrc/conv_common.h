// Shared argument block and output epilogue of the implicit-GEMM conv kernels (fp32 and bf16 MFMA).
#pragma once
#include "bm_common.h"

struct ConvNNArgs {
    const float* x;       long x_bstride;      // [B][Cin][T]
    const float* wp;                           // packed [G][nchunk][KS][BKC][Mpad]
    const int* widx;                           // [B] weight-group per segment, or null (group 0)
    const float* bias;                         // [M] or null
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

// Epilogue for a [32*MT] x [128] tile held as MT 32x32 MFMA accumulators per wavefront (4 wavefronts,
// wavefront w owns columns [32w, 32w+32)): bias, optional pre-activation store, optional per-tile
// BatchNorm partial statistics, optional per-channel affine, activation, residual.
// C/D layout of the 32x32 MFMA (dtype independent): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// `tid` is the thread index inside the 4-wavefront group that owns the tile (a workgroup may hold
// two such groups, one per segment); `smem` is that group's reduction scratch.
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
    float* red = smem;   // [4 waves][BM][2] for the BatchNorm partial statistics (LDS is free now)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int row = m0 + rl;
            const bool ok = col_ok && row < a.M;
            float v = acc[mt][r];
            if (a.bias && row < a.M) v += a.bias[row];
            const long off = (long)row * a.T + col;
            if (a.y_pre && ok) a.y_pre[(long)b * a.y_bstride + off] = v;
            if (a.stats) {
                float s = ok ? v : 0.f;
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
            if (a.y_out && ok) {
                float z = v;
                if (a.ep_scale) z = z * a.ep_scale[row] + a.ep_shift[row];
                z = bm_act(z, a.act, a.leak);
                if (a.res) z += a.res[(long)b * a.res_bstride + off];
                a.y_out[(long)b * a.y_bstride + off] = z;
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
