// Wide-tile time-contraction GEMM in compute mode "f16x2" (see conv_nn_h2w.hip for the arithmetic): the
// weight gradients of the conv stack and of the 1x1 layers,
//
//   KS = 3:  part[split][m][c*3 + j] = sum_{s, t in split} A[s][m][t] * X[s][c][t + (j - 1) * dil]
//   KS = 1:  part[split][m][c]       = sum_{s, t in split} A[s][m][t] * X[s][c][t]
//
// both operands activations: each is scaled by a power of two taken from its tensor maximum (a_amax / x_amax,
// device scalars), split into two f16 planes while it is staged, and the partial tile is multiplied by the
// exact inverse scales on the way out.  Three MFMAs per 32x32x16 block (lo*hi, hi*hi, hi*lo).
//
// Structure = gemm_nt_x3w.hip: ONE workgroup of four wavefronts per CU (one per SIMD); a wavefront owns 160 rows x
// (32 columns x 3 "slots") as 5 x 3 MFMA accumulators; the three slots are the three taps of 64 X rows (KS = 3)
// or three blocks of 64 X rows (KS = 1: workgroup tile 320 x 192).  Stage = 16 samples: 8 pieces of 4 samples
// per thread (A rows 64 i + row64, i < 5, then the three slot copies of the X rows), fetched three stages
// ahead with bounds-checked dwordx4 buffer loads, split and written into the next LDS buffers while the 45 MFMAs
// of the current stage run.  Rows past M / Cn are addressed
// through the per-lane offset, which the buffer descriptor range-checks: they read as zeros and are never
// written, so M and Cn may be padded.
// LDS (16-byte slots = 8 samples of one plane), three stage buffers of two planes each:
//   A [2 halves of the 16 samples][320 + 8 rows], X [3 slots][2 halves][64 + 8 rows]   (102 KB in all).
#include <cstdlib>
#include <cstring>
#include "bm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define HG_BM 320
#define HG_K 16
#define HG_AQ (HG_BM + 8)                     // slots of one 8-sample half of the A tile (8 pad rows)
#define HG_XQ (64 + 8)                        // ... of one slot of the X tile
#define HG_ASLOTS (2 * HG_AQ)                 // 16-byte slots of one plane of the A tile
#define HG_XSLOTS (3 * 2 * HG_XQ)             // ... of the X tile (3 slots)
#define HG_PLANE (HG_ASLOTS + HG_XSLOTS)
#define HG_BUF (2 * HG_PLANE)                 // slots of one stage buffer (2 planes)

struct GemmNTArgsH {
    const float* a; long a_sstride; long a_rstride;
    const float* x; long x_sstride; long x_rstride;
    const float* a_amax; const float* x_amax;
    float* part;
    int S, M, Cn, T, dil, nsplit;
    int tiles_m, tiles_c;
};

// same scale rule as conv_nn_h2w.hip: power of two s with amax * s in [2^14, 2^15), exact inverse
__device__ __forceinline__ void hg_scale_from_amax(float amax, float& s, float& inv) {
    const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;
    int se = 127;
    if (e != 0u && e != 255u) {
        se = 268 - (int)e;
        se = se > 253 ? 253 : (se < 1 ? 1 : se);
    }
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t hg_rsrc(const float* p, int bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Staging loads are compiler-visible raw buffer loads (hipcc counts them and places exact vmcnt(N) waits at
// the first use of each register set).  gemm_nt_x3w.hip hides its loads in inline asm with hand-counted waits;
// that is only sound while the register allocator never copies a register whose load is still in flight, which
// it does here as soon as the schedule changes (observed: v_mov of a staging set ahead of its wait).
__device__ __forceinline__ u32x4 hg_ld128(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
__device__ __forceinline__ unsigned hg_ld32(__amdgpu_buffer_rsrc_t rs, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0);
}

// 4 fp32 values -> scaled f16 planes (hi, lo); one 8-byte LDS store per plane
__device__ __forceinline__ void hg_split_store4(const float (&f)[4], float s, char* dst) {
    f16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xs = f[i] * s;
        const _Float16 a = (_Float16)xs;
        h[i] = a;
        l[i] = (_Float16)(xs - (float)a);
    }
    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2*>(dst + HG_PLANE * 16) = __builtin_bit_cast(u32x2, l);
}

// lgkmcnt(0) + workgroup barrier; names A.hi / B.hi as operands so that the register-only MFMAs that read them
// stay on their side of the barrier (see conv_nn_h2w.hip)
__device__ __forceinline__ void hg_barrier(f16x8 (&ah)[5], f16x8 (&bh)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                 : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2])
                 :: "memory");
}

template <int KS>
__global__ __launch_bounds__(256, 1) void gemm_nt_h2w_kernel(GemmNTArgsH a) {
    constexpr int MW = 5;
    constexpr int BC = KS == 3 ? 64 : 192;            // X rows of the workgroup tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);      // [3 buffers][2 planes][A slots | X slots]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wc = wave & 1;
    const int nl = lane & 31, h = lane >> 5;

    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % a.tiles_m; id /= a.tiles_m;
    const int tc = id % a.tiles_c;
    const int split = id / a.tiles_c;
    const int m0 = tm * HG_BM, c0 = tc * BC;

    const int cps = (a.T + HG_K - 1) / HG_K;
    const long nchunks = (long)a.S * cps;
    const long q_begin = nchunks * split / a.nsplit;
    const long q_end = nchunks * (split + 1) / a.nsplit;
    const int nst = (int)(q_end - q_begin);
    const int halo = KS == 3 ? a.dil : 0;
    const int a_bytes = (int)(((long)(a.M - 1) * a.a_rstride + a.T) * 4);
    const int x_bytes = (int)(((long)(a.Cn - 1) * a.x_rstride + a.T) * 4);
    float sa, sa_inv, sx, sx_inv;
    hg_scale_from_amax(*a.a_amax, sa, sa_inv);
    hg_scale_from_amax(*a.x_amax, sx, sx_inv);

    // The eight 4-sample pieces of this thread: lane -> (row64 = tid >> 2, quarter pq = tid & 3 of the 16
    // samples).  Piece i < 5: A row m0 + 64 i + row64; piece 5 + j: X row c0 + row64 read at tap shift
    // (j - 1) * dil (KS = 3) or X row c0 + 64 j + row64 (KS = 1).  Byte offsets inside a segment, all >= 0; a row
    // past M / Cn gets 0x7f000000: past the range of every descriptor (segments span < 0x7f000000 bytes, checked
    // by the host) and small enough that adding a chunk offset cannot wrap (reads 0).
    const int row64 = tid >> 2, pq = tid & 3;
    int offa[5], offx[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int m = m0 + 64 * i + row64;
        offa[i] = m < a.M ? (m * (int)a.a_rstride + 4 * pq) * 4 : 0x7f000000;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = c0 + (KS == 3 ? 0 : 64 * j) + row64;
        offx[j] = c < a.Cn ? (c * (int)a.x_rstride + 4 * pq) * 4 : 0x7f000000;
    }
    // LDS byte address inside a plane: 16-byte slot of the 8-sample half (pq >> 1), 8-byte half (pq & 1)
    const int ldsa = ((pq >> 1) * HG_AQ + row64) * 16 + (pq & 1) * 8;                 // piece i: + i * 64 * 16
    const int ldsx = (HG_ASLOTS + (pq >> 1) * HG_XQ + row64) * 16 + (pq & 1) * 8;     // slot j: + j * 2 * HG_XQ * 16

    f32x16 acc[MW][3];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[8], rb[8], rc[8];                         // three staging register sets (chunks k+1 .. k+3)

    int ld_q = 0;
    int ld_s = (int)(q_begin / cps);
    int ld_c = (int)(q_begin - (long)ld_s * cps);
    // segment descriptors, rebuilt only when the cursor enters a new segment
    __amdgpu_buffer_rsrc_t qa = hg_rsrc(a.a + (long)ld_s * a.a_sstride, a_bytes);
    __amdgpu_buffer_rsrc_t qx = hg_rsrc(a.x + (long)ld_s * a.x_sstride, x_bytes);

    // loads chunk (ld_s, ld_c) into register set R_ and advances the cursor.  Interior chunks: one dwordx4 per
    // piece (X first: it is consumed a stage earlier).  Edge chunks (first / last of a segment, wave-uniform): one
    // dword per sample, samples whose time index is outside [0, T) get an out-of-range offset and read as 0.
#define HG_LOAD(R_)                                                                               \
    {                                                                                             \
        const int t0 = ld_c * HG_K;                                                               \
        if (!(t0 - halo < 0 || t0 + HG_K + halo > a.T)) {                                         \
            const int s0 = __builtin_amdgcn_readfirstlane(t0 * 4);                                \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)         /* t0 + shift >= 0 in interior chunks */ \
                R_[5 + j] = hg_ld128(qx, offx[j],                                                 \
                                     KS == 3 ? __builtin_amdgcn_readfirstlane((t0 + (j - 1) * a.dil) * 4) : s0); \
            _Pragma("unroll") for (int i = 0; i < 5; ++i) R_[i] = hg_ld128(qa, offa[i], s0);      \
        } else {                                                                                  \
            _Pragma("unroll") for (int ii = 0; ii < 8; ++ii) {                                    \
                const int i = ii < 3 ? 5 + ii : ii - 3;                                           \
                const __amdgpu_buffer_rsrc_t rs = i < 5 ? qa : qx;                                \
                const int shift = (i < 5 || KS != 3) ? 0 : (i - 6) * a.dil;                       \
                const int tt0 = t0 + 4 * pq + shift;                                              \
                const int o = (i < 5 ? offa[i < 5 ? i : 0] : offx[i < 5 ? 0 : i - 5]) + (t0 + shift) * 4; \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
                    const int tt = tt0 + r;                                                       \
                    const int oo = (tt >= 0 && tt < a.T) ? o + r * 4 : 0x7ffffff0;                \
                    R_[i][r] = hg_ld32(rs, oo);                                                   \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
        if (++ld_q < nst) {                            /* else: stays on the last chunk */        \
            if (++ld_c == cps) {                                                                  \
                ld_c = 0; ++ld_s;                                                                 \
                qa = hg_rsrc(a.a + (long)ld_s * a.a_sstride, a_bytes);                            \
                qx = hg_rsrc(a.x + (long)ld_s * a.x_sstride, x_bytes);                            \
            }                                                                                     \
        }                                                                                         \
    }
#define HG_WAIT(R_, N_)       /* waits are the compiler's */
    // splits piece I_ of register set R_ into buffer BUF_ (two 8-byte LDS writes, immediate offsets)
#define HG_STORE(R_, I_, BUF_)                                                                    \
    {                                                                                             \
        char* dst_ = reinterpret_cast<char*>(lds + (BUF_) * HG_BUF) +                             \
                     ((I_) < 5 ? ldsa + (I_) * 64 * 16 : ldsx + ((I_) - 5) * 2 * HG_XQ * 16);     \
        float f_[4];                                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) f_[r] = __uint_as_float(R_[I_][r]);         \
        hg_split_store4(f_, (I_) < 5 ? sa : sx, dst_);                                            \
    }
#define HG_TERM(PA_, PB_)                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                             \
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA_][mt], bf[PB_][j], acc[mt][j], 0, 0, 0);
    // Operand fragments live across stages: [plane 0 = hi, 1 = lo].
    f16x8 af[2][MW], bf[2][3];
    // fragments a stage needs first (A.lo, B.hi), read from buffer BUF_ as soon as it is complete
#define HG_FRAGS_EARLY(BUF_)                                                                      \
    {                                                                                             \
        const u32x4* pb = lds + (BUF_) * HG_BUF;                                                  \
        const u32x4* ab = pb + h * HG_AQ + wm * (MW * 32) + nl;                                   \
        const u32x4* xb = pb + HG_ASLOTS + h * HG_XQ + wc * 32 + nl;                              \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) bf[0][j] = __builtin_bit_cast(f16x8, xb[j * 2 * HG_XQ]); \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[1][mt] = __builtin_bit_cast(f16x8, ab[HG_PLANE + mt * 32]); \
    }
    // One stage k.  On entry A.lo, B.hi of buffer BUF_ are already in registers (read right after the barrier of
    // stage k - 1, under its last 15 MFMAs).  Pieces 0-4 (A) of chunk k + 1 (register set ST_) are split into
    // buffer NBUF_ between the first 30 MFMAs (lo*hi, hi*hi); then the barrier, the early fragments of stage
    // k + 1, pieces 5-7 (X) of chunk k + 2 (set S2_) into buffer N2BUF_ between the last 15 MFMAs (hi*lo), and the
    // fetch of chunk k + 3 into register set LD_ (which held chunk k).  Every stage does all of it, the last ones
    // on clamped (repeated) chunks whose results are never read.
#define HG_STAGE(BUF_, NBUF_, N2BUF_, LD_, ST_, S2_)                                              \
    {                                                                                             \
        const u32x4* pb = lds + (BUF_) * HG_BUF;                                                  \
        const u32x4* ab = pb + h * HG_AQ + wm * (MW * 32) + nl;                                   \
        const u32x4* xb = pb + HG_ASLOTS + h * HG_XQ + wc * 32 + nl;                              \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[0][mt] = __builtin_bit_cast(f16x8, ab[mt * 32]); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) bf[1][j] = __builtin_bit_cast(f16x8, xb[HG_PLANE + j * 2 * HG_XQ]); \
        HG_STORE(ST_, 0, NBUF_)                                                                   \
        HG_STORE(ST_, 1, NBUF_)                                                                   \
        HG_TERM(1, 0)                                                                             \
        HG_STORE(ST_, 2, NBUF_)                                                                   \
        HG_STORE(ST_, 3, NBUF_)                                                                   \
        HG_STORE(ST_, 4, NBUF_)                                                                   \
        HG_TERM(0, 0)                                                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 30; ++g_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        }                                                                                         \
        hg_barrier(af[0], bf[0]);                                                                 \
        HG_FRAGS_EARLY(NBUF_)                                                                     \
        HG_STORE(S2_, 5, N2BUF_)      /* its X pieces go under the last 15 MFMAs */               \
        HG_STORE(S2_, 6, N2BUF_)                                                                  \
        HG_STORE(S2_, 7, N2BUF_)                                                                  \
        HG_TERM(0, 1)                                                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 15; ++g_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        }                                                                                         \
        HG_LOAD(LD_)      /* after the MFMAs: its interior / edge branch ends the scheduling region */ \
    }

    if (nst > 0) {
        HG_LOAD(ra)
        HG_LOAD(rb)
        HG_LOAD(rc)
        HG_STORE(ra, 0, 0) HG_STORE(ra, 1, 0) HG_STORE(ra, 2, 0) HG_STORE(ra, 3, 0)
        HG_STORE(ra, 4, 0) HG_STORE(ra, 5, 0) HG_STORE(ra, 6, 0) HG_STORE(ra, 7, 0)
        HG_STORE(rb, 5, 1) HG_STORE(rb, 6, 1) HG_STORE(rb, 7, 1)
        __syncthreads();
        HG_FRAGS_EARLY(0)
        // stage k reads buffer k % 3; register set k % 3 held chunk k and now receives chunk k + 3.  Pieces 0-4
        // of chunk k + 1 are split before the barrier of stage k, its pieces 5-7 were split after the barrier
        // of stage k - 1 (into a buffer nobody reads before barrier k).
        for (int k = 0; k < nst; k += 3) {
            HG_STAGE(0, 1, 2, ra, rb, rc)
            if (k + 1 < nst) HG_STAGE(1, 2, 0, rb, rc, ra)
            if (k + 2 < nst) HG_STAGE(2, 0, 1, rc, ra, rb)
        }
    }
#undef HG_LOAD
#undef HG_FRAGS_EARLY
#undef HG_WAIT
#undef HG_STORE
#undef HG_TERM
#undef HG_STAGE

    // partial tile out (inverse scales are exact powers of two): KS = 3: part[split][m][c * 3 + j];
    // KS = 1: part[split][m][c0 + 64 j + ...]
    const float f = sa_inv * sx_inv;
    const long N = (long)a.Cn * KS;
    float* dst = a.part + (long)split * a.M * N;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (MW * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < a.M) {
                if (KS == 3) {
                    const int c = c0 + wc * 32 + nl;
                    if (c < a.Cn) {
                        float* p = dst + (long)m * N + (long)c * 3;
                        p[0] = acc[mt][0][r] * f;
                        p[1] = acc[mt][1][r] * f;
                        p[2] = acc[mt][2][r] * f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int c = c0 + 64 * j + wc * 32 + nl;
                        if (c < a.Cn) dst[(long)m * N + c] = acc[mt][j][r] * f;
                    }
                }
            }
        }
    }
}

static bool hg_covers(int S, int G, int M, int Cn, int T, int KS, int dil, bool ordered) {
    static int wide = -1;
    if (wide < 0) {
        const char* e = getenv("BM_H2_WIDE");          // BM_H2_WIDE=0: A/B runs against the 3 x bf16 kernels
        wide = !(e && e[0] == '0');
    }
    if (!wide || G != 1 || ordered || (KS != 3 && KS != 1)) return false;
    if (dil < 1 || dil > 32 || T < 2 * HG_K) return false;
    const int bc = KS == 3 ? 64 : 192;
    // padded rows / columns are wasted MFMA work: at most 25 % (3 taps) / 50 % (1x1 layers, small in absolute terms)
    const long padded = (long)cdiv(M, HG_BM) * HG_BM * cdiv(Cn, bc) * bc;
    if (KS == 3 ? padded * 4 > (long)M * Cn * 5 : padded > (long)M * Cn * 2) return false;
    return (long)S * ((T + HG_K - 1) / HG_K) >= 64;
}

extern "C" int bm_gemm_nt_h2_covers(int M, int Cn, int KS, int S, int T, int G, int dil, int ordered) {
    return hg_covers(S, G, M, Cn, T, KS, dil, ordered != 0) ? 1 : 0;
}

// one workgroup per CU per round (256 CUs), >= 32 stages per workgroup
extern "C" int bm_gemm_nt_h2_suggest_splits(int M, int Cn, int KS, int S, int T) {
    const int tiles = cdiv(M, HG_BM) * cdiv(Cn, KS == 3 ? 64 : 192);
    const long chunks = (long)S * ((T + HG_K - 1) / HG_K);
    long want = 256 / tiles;
    if (want > chunks / 32) want = chunks / 32;
    if (want < 1) want = 1;
    return (int)want;
}

template <int KS>
static int launch_gemm_nt_h2w(const GemmNTArgsH& g, hipStream_t stream) {
    const size_t lds = (size_t)3 * HG_BUF * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_h2w_kernel<KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "gemm_nt_h2w: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long nblocks = (long)g.tiles_m * g.tiles_c * g.nsplit;
    hipLaunchKernelGGL(gemm_nt_h2w_kernel<KS>, dim3((unsigned)nblocks), dim3(256), lds, stream, g);
    return bm_check_launch("gemm_nt_h2w");
}

// part[split][m][c*KS + j] for one group of S consecutive segments (same contract as bm_gemm_nt without
// order / seg); a_amax / x_amax: device pointers to max|a|, max|x| (bm_amax).  Only shapes
// bm_gemm_nt_h2_covers() accepts.
extern "C" int bm_gemm_nt_h2(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* x,
                             long x_sstride, long x_rstride, const float* x_amax, float* part, int S, int M, int Cn,
                             int T, int KS, int dil, int nsplit, void* stream) {
    BM_REQUIRE(a && x && part && a_amax && x_amax, "gemm_nt_h2: null pointer");
    BM_REQUIRE(M > 0 && Cn > 0 && T > 0 && nsplit > 0 && S >= 0, "gemm_nt_h2: bad dims");
    BM_REQUIRE(hg_covers(S, 1, M, Cn, T, KS, dil, false), "gemm_nt_h2: shape not covered (M=%d Cn=%d T=%d KS=%d dil=%d)",
               M, Cn, T, KS, dil);
    BM_REQUIRE((long)M * a_rstride * 4 < 0x7f000000L && (long)Cn * x_rstride * 4 < 0x7f000000L,
               "gemm_nt_h2: a segment spans 2 GB or more");
    GemmNTArgsH g;
    g.a = a; g.a_sstride = a_sstride; g.a_rstride = a_rstride; g.a_amax = a_amax;
    g.x = x; g.x_sstride = x_sstride; g.x_rstride = x_rstride; g.x_amax = x_amax;
    g.part = part; g.S = S; g.M = M; g.Cn = Cn; g.T = T; g.dil = dil; g.nsplit = nsplit;
    g.tiles_m = cdiv(M, HG_BM); g.tiles_c = cdiv(Cn, KS == 3 ? 64 : 192);
    return KS == 3 ? launch_gemm_nt_h2w<3>(g, (hipStream_t)stream) : launch_gemm_nt_h2w<1>(g, (hipStream_t)stream);
}
