// "NT" contraction over time with optional tap shifts, on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
//
//   part[g, split][m][c*KS + j] = sum_{s in group g (split's share)} sum_t
//                                   A[s][m][t] * X[s][c][t + (j - KS/2)*dil]      (zero outside [0,T))
//
// Both operands are time-contiguous, the reduction runs over (segment, time).  It covers:
//   * conv weight gradients  dW[o][c][j] = sum_{b,t} dY[b][o][t] x[b][c][t+shift]   (autograd of common.py:113-114)
//   * per-subject / per-layout grouped weight gradients (SubjectLayers, ChannelMerger; groups via order/seg)
//   * ClipLoss scores  est[B][F*T] . cand[B'][F*T]^T                               (bm/losses.py:94)
//   * ChannelMerger logits heads[O][D] . emb[u][C][D]^T                            (bm/models/common.py:355)
//
// Tiling: a workgroup is WM x WC wavefronts; wavefront (wm, wc) owns MT x NT 32x32 MFMA blocks for
// each of the KS taps.  A rows and X rows are staged in LDS with an odd pitch (conflict-free
// row-per-lane ds_read_b32); the X window carries a +-halo so all taps read the same LDS rows.
// The reduction domain is a flat list of (segment, 32-sample time chunk) pairs; `nsplit` workgroups
// share one output tile (split-K) and write separate partial tiles that pack.hip's
// bm_reduce_splits folds in a fixed order (deterministic).
#include "bm_common.h"

#define BKT 32

struct GemmNTArgs {
    const float* a; long a_sstride; long a_rstride;   // A[s][m][t]
    const float* x; long x_sstride; long x_rstride;   // X[s][c][t]
    const int* order;                                  // segment list (grouped) or null = identity
    const int* seg;                                    // [G+1] group boundaries or null = one group [0, S)
    float* part;                                       // [G*nsplit][M][Cn*KS]
    int S, M, Cn, T, dil, nsplit, G;
    int tiles_m, tiles_c;
};

template <int N> struct FVec { typedef float type __attribute__((ext_vector_type(N))); };

// XWP = padded width of the staged X window (32 + 2*halo <= XWP), compile time so that the staging
// registers can be a fixed-size vector.  The window is centred: it starts HP = (XWP-32)/2 samples
// before the chunk, a multiple of 4, so that with VEC (T % 4 == 0, 16-byte aligned rows) every
// global access is a dwordx4 that lies entirely inside or outside [0, T).
template <int WM, int WC, int MT, int NT, int KS, int XWP, bool VEC>
__global__ __launch_bounds__(WM * WC * 64) void gemm_nt_kernel(GemmNTArgs a) {
    constexpr int NW = WM * WC;
    constexpr int NTH = NW * 64;
    constexpr int BM = WM * MT * 32;
    constexpr int BC = WC * NT * 32;
    constexpr int PA = BKT + 1;
    constexpr int PX = XWP + 1;                     // odd pitch -> conflict-free row-per-lane reads
    constexpr int AVN = BM * BKT / NTH;             // staged A floats per thread
    constexpr int XVN = (BC * XWP + NTH - 1) / NTH; // staged X floats per thread
    static_assert(BM * BKT % NTH == 0, "A tile must split evenly");
    static_assert(!VEC || (AVN % 4 == 0 && (BC * XWP) % (4 * NTH) == 0), "vector staging must split evenly");
    typedef typename FVec<AVN>::type avec_t;
    typedef typename FVec<XVN>::type xvec_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WC, wc = wave % WC;
    const int nl = lane & 31, h = lane >> 5;

    constexpr int HP = (XWP - BKT) / 2;        // window start = t0 - HP
    const int halo = (KS >> 1) * a.dil;
    const int xoff = HP - halo;                // column of tap 0 inside the window
    float* As = smem;                          // [BM][PA]
    float* Xs = smem + BM * PA;                // [BC][PX]

    // block -> (tile_m, tile_c, split, g).  XCD-aware: all tiles of one (g, split) -- which stream
    // the SAME segments -- get consecutive logical ids, i.e. run on one XCD and share its L2.
    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % a.tiles_m; id /= a.tiles_m;
    const int tc = id % a.tiles_c; id /= a.tiles_c;
    const int split = id % a.nsplit;
    const int g = id / a.nsplit;
    const int m0 = tm * BM, c0 = tc * BC;

    const int s_begin = a.seg ? a.seg[g] : 0;
    const int s_end = a.seg ? a.seg[g + 1] : a.S;
    const int cps = (a.T + BKT - 1) / BKT;                     // chunks per segment
    const long nchunks = (long)(s_end - s_begin) * cps;
    const long q_begin = nchunks * split / a.nsplit;
    const long q_end = nchunks * (split + 1) / a.nsplit;

    f32x16 acc[MT][NT][KS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][k][j][r] = 0.f;

    avec_t areg;
    xvec_t xreg;
    int t0 = 0;

    // Issue the global loads of chunk q into registers (no wait).
#define NT_LOAD(Q_)                                                                               \
    {                                                                                             \
        const int sl = (int)((Q_) / cps);                                                         \
        t0 = (int)((Q_) - (long)sl * cps) * BKT;                                                  \
        int sidx = s_begin + sl;                                                                  \
        if (a.order) sidx = a.order[sidx];                                                        \
        const float* ab = a.a + (long)sidx * a.a_sstride;                                         \
        const float* xb = a.x + (long)sidx * a.x_sstride;                                         \
        if constexpr (VEC) {                                                                      \
            _Pragma("unroll") for (int k = 0; k < AVN / 4; ++k) {                                 \
                const int e = tid + k * NTH;                                                      \
                const int i = e >> 3, q = e & 7;                                                  \
                const int m = m0 + i, t = t0 + 4 * q;                                             \
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
                if (m < a.M && t < a.T)                                                           \
                    v = *reinterpret_cast<const float4*>(ab + (long)m * a.a_rstride + t);         \
                areg[4 * k] = v.x; areg[4 * k + 1] = v.y; areg[4 * k + 2] = v.z; areg[4 * k + 3] = v.w; \
            }                                                                                     \
            _Pragma("unroll") for (int k = 0; k < XVN / 4; ++k) {                                 \
                const int e = tid + k * NTH;                                                      \
                const int i = e / (XWP / 4), q = e - i * (XWP / 4);                               \
                const int c = c0 + i, t = t0 - HP + 4 * q;                                        \
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
                if (i < BC && c < a.Cn && t >= 0 && t < a.T)                                      \
                    v = *reinterpret_cast<const float4*>(xb + (long)c * a.x_rstride + t);         \
                xreg[4 * k] = v.x; xreg[4 * k + 1] = v.y; xreg[4 * k + 2] = v.z; xreg[4 * k + 3] = v.w; \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int k = 0; k < AVN; ++k) {                                     \
                const int i = (tid >> 5) + k * (NTH / 32);                                        \
                const int m = m0 + i, t = t0 + (tid & 31);                                        \
                areg[k] = (m < a.M && t < a.T) ? ab[(long)m * a.a_rstride + t] : 0.f;             \
            }                                                                                     \
            _Pragma("unroll") for (int k = 0; k < XVN; ++k) {                                     \
                const int e = tid + k * NTH;                                                      \
                const int i = e / XWP, xx = e - i * XWP;                                          \
                const int c = c0 + i, t = t0 - HP + xx;                                           \
                xreg[k] = (i < BC && c < a.Cn && t >= 0 && t < a.T) ? xb[(long)c * a.x_rstride + t] : 0.f; \
            }                                                                                     \
        }                                                                                         \
    }
#define NT_STORE()                                                                                \
    {                                                                                             \
        if constexpr (VEC) {                                                                      \
            _Pragma("unroll") for (int k = 0; k < AVN / 4; ++k) {                                 \
                const int e = tid + k * NTH;                                                      \
                float* d = As + (e >> 3) * PA + 4 * (e & 7);                                      \
                d[0] = areg[4 * k]; d[1] = areg[4 * k + 1]; d[2] = areg[4 * k + 2]; d[3] = areg[4 * k + 3]; \
            }                                                                                     \
            _Pragma("unroll") for (int k = 0; k < XVN / 4; ++k) {                                 \
                const int e = tid + k * NTH;                                                      \
                const int i = e / (XWP / 4), q = e - i * (XWP / 4);                               \
                if (i < BC) {                                                                     \
                    float* d = Xs + i * PX + 4 * q;                                               \
                    d[0] = xreg[4 * k]; d[1] = xreg[4 * k + 1]; d[2] = xreg[4 * k + 2]; d[3] = xreg[4 * k + 3]; \
                }                                                                                 \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int k = 0; k < AVN; ++k)                                       \
                As[((tid >> 5) + k * (NTH / 32)) * PA + (tid & 31)] = areg[k];                    \
            _Pragma("unroll") for (int k = 0; k < XVN; ++k) {                                     \
                const int e = tid + k * NTH;                                                      \
                const int i = e / XWP, xx = e - i * XWP;                                          \
                if (i < BC) Xs[i * PX + xx] = xreg[k];                                            \
            }                                                                                     \
        }                                                                                         \
    }

    if (q_begin < q_end) {
        NT_LOAD(q_begin);
        NT_STORE();
    }
    __syncthreads();
    for (long q = q_begin; q < q_end; ++q) {
        const int tvalid = min(BKT, a.T - t0);
        const int ksteps = (tvalid + 1) >> 1;
        const bool more = q + 1 < q_end;
        if (more) NT_LOAD(q + 1);              // in flight during the MFMAs below (t0 now = next chunk)
        const float* ap = As + (wm * MT * 32 + nl) * PA + h;
        const float* xp = Xs + (wc * NT * 32 + nl) * PX + h;
#define NT_KSTEP(KK_)                                                                              \
    {                                                                                             \
        float av[MT];                                                                             \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) av[i] = ap[i * 32 * PA + 2 * (KK_)];       \
        _Pragma("unroll") for (int k = 0; k < NT; ++k) {                                          \
            _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                      \
                const float bv = xp[k * 32 * PX + 2 * (KK_) + xoff + j * a.dil];                         \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i][k][j] =                     \
                    __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i][k][j], 0, 0, 0);       \
            }                                                                                     \
        }                                                                                         \
    }
        // groups of 4 k-steps, fully unrolled inside: hipcc hoists the 4*(MT + NT*KS) LDS operand
        // reads of a group over its MFMAs; a partial last chunk only runs the groups it needs
        // (the LDS tile is zero-filled beyond T).
        const int kgroups = (ksteps + 3) >> 2;
        for (int g4 = 0; g4 < kgroups; ++g4) {
            NT_KSTEP(4 * g4 + 0)
            NT_KSTEP(4 * g4 + 1)
            NT_KSTEP(4 * g4 + 2)
            NT_KSTEP(4 * g4 + 3)
        }
#undef NT_KSTEP
        __syncthreads();
        if (more) {
            NT_STORE();
            __syncthreads();
        }
    }
#undef NT_LOAD
#undef NT_STORE

    // epilogue: part[(g*nsplit+split)][m][c*KS + j]
    const long N = (long)a.Cn * KS;
    float* dst = a.part + (long)(g * a.nsplit + split) * a.M * N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int c = c0 + wc * NT * 32 + k * 32 + nl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < a.M && c < a.Cn) {
#pragma unroll
                    for (int j = 0; j < KS; ++j) dst[(long)m * N + (long)c * KS + j] = acc[i][k][j][r];
                }
            }
        }
}

template <int WM, int WC, int MT, int NT, int KS, int XWP, bool VEC>
static int launch_gemm_nt_v(GemmNTArgs a, hipStream_t stream) {
    constexpr int BM = WM * MT * 32, BC = WC * NT * 32;
    const size_t lds = (size_t)(BM * (BKT + 1) + BC * (XWP + 1)) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<WM, WC, MT, NT, KS, XWP, VEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "gemm_nt: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_c = cdiv(a.Cn, BC);
    const long nblocks = (long)a.tiles_m * a.tiles_c * a.nsplit * a.G;
    if (nblocks <= 0) return BM_OK;
    hipLaunchKernelGGL((gemm_nt_kernel<WM, WC, MT, NT, KS, XWP, VEC>), dim3((unsigned)nblocks),
                       dim3(WM * WC * 64), lds, stream, a);
    return bm_check_launch("gemm_nt");
}

template <int WM, int WC, int MT, int NT, int KS, int XWP>
static int launch_gemm_nt_w(GemmNTArgs a, hipStream_t stream) {
    // dwordx4 staging needs every row start 16-byte aligned and T a multiple of 4
    const bool vec = (a.T % 4 == 0) && (a.a_rstride % 4 == 0) && (a.x_rstride % 4 == 0) &&
                     (a.a_sstride % 4 == 0) && (a.x_sstride % 4 == 0) &&
                     (((uintptr_t)a.a | (uintptr_t)a.x) % 16 == 0);
    if (vec) return launch_gemm_nt_v<WM, WC, MT, NT, KS, XWP, true>(a, stream);
    return launch_gemm_nt_v<WM, WC, MT, NT, KS, XWP, false>(a, stream);
}

template <int WM, int WC, int MT, int NT, int KS>
static int launch_gemm_nt(GemmNTArgs a, hipStream_t stream) {
    const int halo = (KS >> 1) * a.dil;
    if (KS == 1) return launch_gemm_nt_w<WM, WC, MT, NT, KS, 32>(a, stream);
    if (halo <= 16) return launch_gemm_nt_w<WM, WC, MT, NT, KS, 64>(a, stream);
    if (halo <= 32) return launch_gemm_nt_w<WM, WC, MT, NT, KS, 96>(a, stream);
    return bm_set_error(BM_ERR_UNSUPPORTED, "gemm_nt: (kernel_size/2)*dilation = %d exceeds the 32-sample halo", halo);
}

// Picks the tile (128 or 64 rows / cols) with the least padding.
static inline bool prefer_big(int n) { return (long)cdiv(n, 128) * 128 <= (long)cdiv(n, 64) * 64; }

// Suggested split count so that the launch has ~>= 4 workgroups per CU.
extern "C" int bm_gemm_nt_suggest_splits(int M, int Cn, int KS, int S, int T, int G) {
    int tiles;
    if (KS == 1) tiles = cdiv(M, prefer_big(M) ? 128 : 64) * cdiv(Cn, prefer_big(Cn) ? 128 : 64);
    else tiles = cdiv(M, prefer_big(M) ? 128 : 64) * cdiv(Cn, 64);
    long chunks = (long)S * cdiv(T, BKT);
    if (G > 1) return 1;
    long want = (1024 + tiles - 1) / tiles;
    if (want > chunks / 8) want = chunks / 8;     // keep >= 8 chunks of work per workgroup
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    return (int)want;
}

// Split count of the ClipLoss score contraction (M x Cn outputs, K = S * T long): one workgroup per CU -- more
// splits only add partial-tile traffic (nsplit * M * Cn * 4 bytes written here, read again by bm_clip_ce).
extern "C" int bm_clip_suggest_splits(int M, int Cn, int S, int T) {
    const int tiles = cdiv(M, prefer_big(M) ? 128 : 64) * cdiv(Cn, prefer_big(Cn) ? 128 : 64);
    const long chunks = (long)S * cdiv(T, BKT);
    long want = (256 + tiles - 1) / tiles;
    if (want > chunks / 8) want = chunks / 8;
    if (want < 1) want = 1;
    return (int)want;
}

extern "C" int bm_gemm_nt(const float* a, long a_sstride, long a_rstride, const float* x,
                          long x_sstride, long x_rstride, const int* order, const int* seg,
                          float* part, int S, int G, int M, int Cn, int T, int KS, int dil,
                          int nsplit, void* stream) {
    BM_REQUIRE(a && x && part, "gemm_nt: null pointer");
    BM_REQUIRE(M > 0 && Cn > 0 && T > 0 && G > 0 && nsplit > 0 && S >= 0, "gemm_nt: bad dims");
    BM_REQUIRE(G == 1 || seg, "gemm_nt: grouped call needs seg[]");
    GemmNTArgs g;
    g.a = a; g.a_sstride = a_sstride; g.a_rstride = a_rstride;
    g.x = x; g.x_sstride = x_sstride; g.x_rstride = x_rstride;
    g.order = order; g.seg = seg; g.part = part;
    g.S = S; g.M = M; g.Cn = Cn; g.T = T; g.dil = dil; g.nsplit = nsplit; g.G = G;
    hipStream_t s = (hipStream_t)stream;
    const bool bigM = prefer_big(M);
    if (KS == 1) {
        const bool bigC = prefer_big(Cn);
        if (bigM && bigC) return launch_gemm_nt<2, 2, 2, 2, 1>(g, s);
        if (bigM) return launch_gemm_nt<2, 2, 2, 1, 1>(g, s);
        if (bigC) return launch_gemm_nt<2, 2, 1, 2, 1>(g, s);
        return launch_gemm_nt<2, 2, 1, 1, 1>(g, s);
    }
    if (KS == 3) {
        if (bigM) return launch_gemm_nt<2, 2, 2, 1, 3>(g, s);
        return launch_gemm_nt<2, 2, 1, 1, 3>(g, s);
    }
    if (KS == 5) return launch_gemm_nt<2, 2, 1, 1, 5>(g, s);
    return bm_set_error(BM_ERR_UNSUPPORTED, "gemm_nt: kernel size %d not supported (1, 3, 5)", KS);
}
