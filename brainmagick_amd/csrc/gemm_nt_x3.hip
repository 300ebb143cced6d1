// fp32-ACCURATE time-contraction GEMM on the bf16 matrix cores (compute mode "f32x3", see
// conv_nn_x3.hip): both operands are split exactly into three bf16 planes while they are staged into
// LDS and every 32x32x16 block is evaluated as six bf16 MFMAs (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid,
// hi*hi) with fp32 accumulation.
//
//   part[g, split][m][c*KS + j] = sum_{s in group g} sum_t A[s][m][t] * X[s][c][t + (j - KS/2)*dil]
//
// LDS: rows of 32 samples = 4 slots of 16 bytes, slot index XOR-swizzled with bits 2-3 of the row so that
// both the row-per-lane ds_read_b128 of the MFMA operands and the 8-lane ds_write_b128 groups of the
// staging pass are bank-conflict free (a +1 pad slot made the writes 2-way); three planes per operand, one pre-shifted copy of the X tile per tap (as in
// gemm_nt_bf16.hip).  Software pipeline, split-K, grouping and XCD mapping as in gemm_nt.hip.
#include "bm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BKT2 32
#define SLOTS 4          // 4 data slots per row, XOR-swizzled: slot (row, q) lives at row*4 + (q ^ ((row >> 2) & 3))
#define DSL 4            // data slots per row
#define SWZ(ROW_, Q_) ((ROW_) * SLOTS + ((Q_) ^ (((ROW_) >> 2) & 3)))

struct GemmNTArgsX {
    const float* a; long a_sstride; long a_rstride;
    const float* x; long x_sstride; long x_rstride;
    const int* order;
    const int* seg;
    float* part;
    int S, M, Cn, T, dil, nsplit, G;
    int tiles_m, tiles_c;
};

template <int N> struct FVecD { typedef float type __attribute__((ext_vector_type(N))); };

// exact 3-way split of 8 fp32 values into bf16 planes (hi, mid, lo)
__device__ __forceinline__ void split8x(const float* f, u32x4& hi, u32x4& mid, u32x4& lo) {
    bf16x8 h, m, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)f[i];
        const float r1 = f[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[i] = a; m[i] = b; l[i] = (__bf16)(r1 - (float)b);
    }
    hi = __builtin_bit_cast(u32x4, h);
    mid = __builtin_bit_cast(u32x4, m);
    lo = __builtin_bit_cast(u32x4, l);
}

// wave-uniform buffer descriptor over one segment: out-of-range dwords (rows past the end, t < 0 on the
// first row) read as 0 without any per-lane predicate
__device__ __forceinline__ __amdgpu_buffer_rsrc_t seg_rsrc(const float* p, int bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int WM, int WC, int MT, int NT, int KS>
__global__ __launch_bounds__(WM * WC * 64, (MT * NT * KS <= 3) ? 3 : 2) void gemm_nt_x3_kernel(GemmNTArgsX a) {
    constexpr int NW = WM * WC;
    constexpr int NTH = NW * 64;
    constexpr int BM = WM * MT * 32;
    constexpr int BC = WC * NT * 32;
    constexpr int AIT = BM * DSL / NTH;               // 8-sample items per thread (A)
    constexpr int XPT = BC * DSL / NTH;               // 8-sample items per thread and tap (X)
    constexpr int XIT = KS * XPT;
    static_assert((BM * DSL) % NTH == 0 && (BC * DSL) % NTH == 0, "tiles must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);       // [3 planes][BM][SLOTS]
    u32x4* Xs = As + 3 * BM * SLOTS;                  // [3 planes][KS][BC][SLOTS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WC, wc = wave % WC;
    const int nl = lane & 31, h = lane >> 5;

    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % a.tiles_m; id /= a.tiles_m;
    const int tc = id % a.tiles_c; id /= a.tiles_c;
    const int split = id % a.nsplit;
    const int g = id / a.nsplit;
    const int m0 = tm * BM, c0 = tc * BC;

    const int s_begin = a.seg ? a.seg[g] : 0;
    const int s_end = a.seg ? a.seg[g + 1] : a.S;
    const int cps = (a.T + BKT2 - 1) / BKT2;
    const long nchunks = (long)(s_end - s_begin) * cps;
    const long q_begin = nchunks * split / a.nsplit;
    const long q_end = nchunks * (split + 1) / a.nsplit;
    const int halo = (KS >> 1) * a.dil;
    const int a_bytes = (int)(((long)(a.M - 1) * a.a_rstride + a.T) * 4);
    const int x_bytes = (int)(((long)(a.Cn - 1) * a.x_rstride + a.T) * 4);

    f32x16 acc[MT][NT][KS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][k][j][r] = 0.f;

    typename FVecD<8 * AIT>::type areg;
    typename FVecD<8 * XIT>::type xreg;

    // per-thread element offsets inside a segment (row * stride + 8 * slot), fixed for the whole kernel
    int a_off[AIT], x_off[XPT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int e = tid + i * NTH;
        a_off[i] = (m0 + (e >> 2)) * (int)a.a_rstride + 8 * (e & 3);
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * NTH;
        x_off[i] = (c0 + (e >> 2)) * (int)a.x_rstride + 8 * (e & 3);
    }

    // interior chunks: 8 consecutive samples starting at element offset OFF_ (>= 0) of the segment behind
    // RS_ as two dwordx4 buffer loads (dword alignment is all the hardware asks for, so every tap shift
    // takes this path); rows past the end of the segment come back as 0 from the buffer bounds check
#define LOAD8(DST_, D0_, RS_, OFF_)                                                               \
    {                                                                                             \
        const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(RS_, (OFF_) * 4, 0, 0);            \
        const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(RS_, (OFF_) * 4 + 16, 0, 0);       \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                           \
            DST_[D0_ + r] = __uint_as_float(v0[r]);                                               \
            DST_[D0_ + 4 + r] = __uint_as_float(v1[r]);                                           \
        }                                                                                         \
    }
    // edge chunks (wave-uniform, the first / last chunks of a segment): one dword per sample, samples whose
    // time index falls outside [0, T) get an out-of-range offset and therefore read as 0.  (A negative
    // voffset must never meet a positive immediate offset: the bounds check does not wrap.)
#define LOAD8E(DST_, D0_, RS_, OFF_, T_)                                                          \
    {                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                           \
            const int tt = (T_) + r;                                                              \
            const int off = (tt >= 0 && tt < a.T) ? ((OFF_) + r) * 4 : 0x7ffffff0;                \
            DST_[D0_ + r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(RS_, off, 0, 0)); \
        }                                                                                         \
    }
#define NT_LOAD()        /* stages chunk (ld_s, ld_c) and advances the cursor */                  \
    {                                                                                             \
        const int t0 = ld_c * BKT2;                                                               \
        int sidx = s_begin + ld_s;                                                                \
        if (a.order) sidx = a.order[sidx];                                                        \
        const __amdgpu_buffer_rsrc_t ra = seg_rsrc(a.a + (long)sidx * a.a_sstride, a_bytes);      \
        const __amdgpu_buffer_rsrc_t rx = seg_rsrc(a.x + (long)sidx * a.x_sstride, x_bytes);      \
        if (t0 - halo < 0 || t0 + BKT2 + halo > a.T) {                                            \
            _Pragma("unroll") for (int i = 0; i < AIT; ++i)                                       \
                LOAD8E(areg, 8 * i, ra, a_off[i] + t0, t0 + 8 * ((tid + i * NTH) & 3))            \
            _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                      \
                const int shift = (j - (KS >> 1)) * a.dil;                                        \
                _Pragma("unroll") for (int i = 0; i < XPT; ++i)                                   \
                    LOAD8E(xreg, 8 * (j * XPT + i), rx, x_off[i] + t0 + shift,                    \
                           t0 + shift + 8 * ((tid + i * NTH) & 3))                                \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < AIT; ++i) LOAD8(areg, 8 * i, ra, a_off[i] + t0)  \
            _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                      \
                const int shift = (j - (KS >> 1)) * a.dil;                                        \
                _Pragma("unroll") for (int i = 0; i < XPT; ++i)                                   \
                    LOAD8(xreg, 8 * (j * XPT + i), rx, x_off[i] + t0 + shift)                     \
            }                                                                                     \
        }                                                                                         \
        if (++ld_c == cps) { ld_c = 0; ++ld_s; }                                                  \
    }
#define NT_STORE()                                                                                \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                         \
            const int e = tid + i * NTH;                                                          \
            float f[8];                                                                           \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) f[r] = areg[8 * i + r];                 \
            u32x4 hi, mid, lo;                                                                    \
            split8x(f, hi, mid, lo);                                                              \
            const int slot = SWZ(e >> 2, e & 3);                                                  \
            As[slot] = hi; As[BM * SLOTS + slot] = mid; As[2 * BM * SLOTS + slot] = lo;           \
        }                                                                                         \
        _Pragma("unroll") for (int j = 0; j < KS; ++j)                                            \
            _Pragma("unroll") for (int i = 0; i < XPT; ++i) {                                     \
                const int e = tid + i * NTH;                                                      \
                float f[8];                                                                       \
                _Pragma("unroll") for (int r = 0; r < 8; ++r) f[r] = xreg[8 * (j * XPT + i) + r]; \
                u32x4 hi, mid, lo;                                                                \
                split8x(f, hi, mid, lo);                                                          \
                const int slot = j * BC * SLOTS + SWZ(e >> 2, e & 3);                             \
                Xs[slot] = hi; Xs[KS * BC * SLOTS + slot] = mid; Xs[2 * KS * BC * SLOTS + slot] = lo; \
            }                                                                                     \
    }

    int ld_s = (int)(q_begin / cps);
    int ld_c = (int)(q_begin - (long)ld_s * cps);
    if (q_begin < q_end) {
        NT_LOAD();
        NT_STORE();
    }
    __syncthreads();
    for (long q = q_begin; q < q_end; ++q) {
        const bool more = q + 1 < q_end;
        if (more) NT_LOAD();
        // row-per-lane operand reads; rows of the 32-row MFMA blocks start at multiples of 32, so the
        // swizzle term only depends on the lane's row-in-block nl
        const u32x4* ap = As + (wm * MT * 32 + nl) * SLOTS;
        const u32x4* xp = Xs + (wc * NT * 32 + nl) * SLOTS;
        const int sw = (nl >> 2) & 3;
        constexpr int APL = BM * SLOTS, XPL = KS * BC * SLOTS;      // plane strides
#pragma unroll
        for (int kk = 0; kk < BKT2 / 16; ++kk) {
            bf16x8 ah[MT], am[MT], al[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int q = (2 * kk + h) ^ sw;
                ah[i] = __builtin_bit_cast(bf16x8, ap[i * 32 * SLOTS + q]);
                am[i] = __builtin_bit_cast(bf16x8, ap[APL + i * 32 * SLOTS + q]);
                al[i] = __builtin_bit_cast(bf16x8, ap[2 * APL + i * 32 * SLOTS + q]);
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const int o = (j * BC + k * 32) * SLOTS + ((2 * kk + h) ^ sw);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, xp[o]);
                    const bf16x8 bm = __builtin_bit_cast(bf16x8, xp[XPL + o]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, xp[2 * XPL + o]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        f32x16 c = acc[i][k][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, c, 0, 0, 0);
                        acc[i][k][j] = c;
                    }
                }
            }
        }
        __syncthreads();
        if (more) {
            NT_STORE();
            __syncthreads();
        }
    }
#undef NT_LOAD
#undef NT_STORE
#undef LOAD8
#undef LOAD8E

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

template <int WM, int WC, int MT, int NT, int KS>
static int launch_gemm_nt_x3(GemmNTArgsX a, hipStream_t stream) {
    constexpr int BM = WM * MT * 32, BC = WC * NT * 32;
    const size_t lds = (size_t)3 * (BM + KS * BC) * SLOTS * 16;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_x3_kernel<WM, WC, MT, NT, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "gemm_nt_x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_c = cdiv(a.Cn, BC);
    const long nblocks = (long)a.tiles_m * a.tiles_c * a.nsplit * a.G;
    if (nblocks <= 0) return BM_OK;
    hipLaunchKernelGGL((gemm_nt_x3_kernel<WM, WC, MT, NT, KS>), dim3((unsigned)nblocks),
                       dim3(WM * WC * 64), lds, stream, a);
    return bm_check_launch("gemm_nt_x3");
}

extern "C" int bm_gemm_nt(const float* a, long a_sstride, long a_rstride, const float* x, long x_sstride,
                          long x_rstride, const int* order, const int* seg, float* part, int S, int G, int M,
                          int Cn, int T, int KS, int dil, int nsplit, void* stream);   // gemm_nt.hip

static inline bool prefer_big_x(int n) { return (long)cdiv(n, 128) * 128 <= (long)cdiv(n, 64) * 64; }

// Same contract as bm_gemm_nt (fp32 partial tiles out); fp32-accurate 3-plane bf16 emulation.
extern "C" int bm_gemm_nt_x3(const float* a, long a_sstride, long a_rstride, const float* x,
                               long x_sstride, long x_rstride, const int* order, const int* seg,
                               float* part, int S, int G, int M, int Cn, int T, int KS, int dil,
                               int nsplit, void* stream) {
    BM_REQUIRE(a && x && part, "gemm_nt_x3: null pointer");
    BM_REQUIRE(M > 0 && Cn > 0 && T > 0 && G > 0 && nsplit > 0 && S >= 0, "gemm_nt_x3: bad dims");
    BM_REQUIRE(G == 1 || seg, "gemm_nt_x3: grouped call needs seg[]");
    // The staging addresses are 32-bit byte offsets into one segment (buffer descriptors).  An operand whose
    // segment spans 2 GB or more (e.g. 2 048 wav2vec candidates x 368 640 samples on 8 GPUs) goes through the
    // exact-fp32 kernel, which addresses with 64-bit pointers (same contract, at least as accurate).
    if (((long)(M - 1) * a_rstride + T) * 4 >= 0x7fffff00L || ((long)(Cn - 1) * x_rstride + T) * 4 >= 0x7fffff00L)
        return bm_gemm_nt(a, a_sstride, a_rstride, x, x_sstride, x_rstride, order, seg, part, S, G, M, Cn, T, KS,
                          dil, nsplit, stream);
    GemmNTArgsX g;
    g.a = a; g.a_sstride = a_sstride; g.a_rstride = a_rstride;
    g.x = x; g.x_sstride = x_sstride; g.x_rstride = x_rstride;
    g.order = order; g.seg = seg; g.part = part;
    g.S = S; g.M = M; g.Cn = Cn; g.T = T; g.dil = dil; g.nsplit = nsplit; g.G = G;
    hipStream_t s = (hipStream_t)stream;
    const bool bigM = prefer_big_x(M);
    if (KS == 1) {
        const bool bigC = prefer_big_x(Cn);
        if (bigM && bigC) return launch_gemm_nt_x3<2, 2, 2, 2, 1>(g, s);
        if (bigM) return launch_gemm_nt_x3<2, 2, 2, 1, 1>(g, s);
        if (bigC) return launch_gemm_nt_x3<2, 2, 1, 2, 1>(g, s);
        return launch_gemm_nt_x3<2, 2, 1, 1, 1>(g, s);
    }
    if (KS == 3) {
        if (bigM) return launch_gemm_nt_x3<2, 2, 2, 1, 3>(g, s);
        return launch_gemm_nt_x3<2, 2, 1, 1, 3>(g, s);
    }
    if (KS == 5) return launch_gemm_nt_x3<2, 2, 1, 1, 5>(g, s);
    return bm_set_error(BM_ERR_UNSUPPORTED, "gemm_nt_x3: kernel size %d not supported (1, 3, 5)", KS);
}
