// Wide-tile time-contraction GEMM in compute mode "f16x2" (see conv_nn_h2w.hip for the arithmetic): the weight
// gradients of the conv stack and of the 1x1 layers, and the ClipLoss score contraction,
//
//   KS = 3:  part[split][m][c*3 + j] = sum_{s, t in split} A[s][m][t] * X[s][c][t + (j - 1) * dil]
//   KS = 1:  part[split][m][c]       = sum_{s, t in split} A[s][m][t] * X[s][c][t]
//
// both operands activations: each is scaled by a power of two taken from its tensor maximum (a_amax / x_amax,
// device scalars), split into two f16 planes while it is staged, and the partial tile is multiplied by the
// exact inverse scales on the way out.  Three MFMAs per 32x32x16 block (lo*hi, hi*hi, hi*lo).
//
// ONE workgroup of four wavefronts per CU (one per SIMD); a wavefront owns (32 MW) rows x (32 columns x NS
// "slots") as MW x NS MFMA accumulators; the slots are the three taps of 64 X rows (KS = 3, NS = 3) or NS blocks
// of 64 X rows (KS = 1).  Tiles: <3,5,3> and <1,5,3> 320 x 192 (weight gradients), <1,4,2> 256 x 128 (ClipLoss
// scores at batch 256: two column tiles x 128 K-splits fill the chip with 33 MB of partial tiles).
// Stage = 16 samples = MW + NS pieces of 4 samples per thread (A rows 64 i + row64, then the NS slot copies of the
// X rows).  Pipeline of a stage k (MFMAs on LDS buffer k & 1):
//     wait for chunk k + 1 (fetched a whole stage ago)  ->  fetch chunk k + 2 into the other register set  ->
//     split chunk k + 1 into LDS buffer (k + 1) & 1 between the first 2/3 of the MFMAs (lo*hi, hi*hi)  ->
//     barrier  ->  early fragments of stage k + 1  ->  last third of the MFMAs (hi*lo).
// The loads are compiler-visible raw buffer loads: gemm_nt_x3w.hip hides its loads in inline asm with hand-counted
// waits, which is only sound while the register allocator never copies a register whose load is still in
// flight -- it did here (v_mov of a staging set ahead of its wait).  hipcc's own waits are conservative around
// the interior / edge branch of the fetch, so the schedule gives every fetch a full stage before its first use
// and names all its registers at that point (one wait, nothing left pending when the next fetch is issued).
// Rows past M / Cn are addressed through the per-lane offset, which the buffer descriptor range-checks: they
// read as zeros and are never written, so M and Cn may be padded.
// LDS (16-byte slots = 8 samples of one plane), two stage buffers of two planes each:
//   A [2 halves of the 16 samples][64 MW + 8 rows], X [NS slots][2 halves][64 + 8 rows].
#include <cstdlib>
#include <cstring>
#include "bm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define HG_K 16
#define HG_XQ (64 + 8)                        // slots of one 8-sample half of one X slot (8 pad rows)

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
__device__ __forceinline__ void hg_split_store4(const float (&f)[4], float s, char* dst, int plane_bytes) {
    f16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xs = f[i] * s;
        const _Float16 a = (_Float16)xs;
        h[i] = a;
        l[i] = (_Float16)(xs - (float)a);
    }
    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2*>(dst + plane_bytes) = __builtin_bit_cast(u32x2, l);
}

// lgkmcnt(0) + workgroup barrier; names A.hi / B.hi as operands so that the register-only MFMAs that read them
// stay on their side of the barrier (see conv_nn_h2w.hip)
template <int MW, int NS>
__device__ __forceinline__ void hg_barrier(f16x8 (&ah)[MW], f16x8 (&bh)[NS]) {
    if constexpr (MW == 5 && NS == 3)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2])
                     :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(bh[0]), "+v"(bh[1])
                     :: "memory");
}

// "every register of this staging set is needed now": the compiler places its wait for the whole set here
template <int NP>
__device__ __forceinline__ void hg_touch(u32x4 (&r)[NP]) {
    if constexpr (NP == 8)
        asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
    else
        asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]));
}

template <int KS, int MW, int NS>
__global__ __launch_bounds__(256, 1) void gemm_nt_h2w_kernel(GemmNTArgsH a) {
    static_assert((MW == 5 && NS == 3) || (MW == 4 && NS == 2), "tile variants: 320 x 192, 256 x 128");
    static_assert(KS == 1 || NS == 3, "3 taps use the 3 slots");
    constexpr int NP = MW + NS;                       // 4-sample pieces per thread and stage
    constexpr int BM = 64 * MW;
    constexpr int BC = KS == 3 ? 64 : 64 * NS;        // X rows of the workgroup tile
    constexpr int AQ = BM + 8;                        // slots of one 8-sample half of the A tile (8 pad rows)
    constexpr int ASLOTS = 2 * AQ;                    // 16-byte slots of one plane of the A tile
    constexpr int PLANE = ASLOTS + NS * 2 * HG_XQ;    // ... of one plane (A + X)
    constexpr int BUF = 2 * PLANE;                    // slots of one stage buffer (2 planes)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);      // [2 buffers][2 planes][A slots | X slots]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wc = wave & 1;
    const int nl = lane & 31, h = lane >> 5;

    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % a.tiles_m; id /= a.tiles_m;
    const int tc = id % a.tiles_c;
    const int split = id / a.tiles_c;
    const int m0 = tm * BM, c0 = tc * BC;

    const int cps = (a.T + HG_K - 1) / HG_K;
    const long nchunks = (long)a.S * cps;
    const long q_begin = nchunks * split / a.nsplit;
    const long q_end = nchunks * (split + 1) / a.nsplit;
    const int nst = (int)(q_end - q_begin);
    const int halo = KS == 3 ? a.dil : 0;
    const int a_bytes = (int)(((long)(a.M - 1) * a.a_rstride + a.T) * 4);
    const int x_bytes = (int)(((long)(a.Cn - 1) * a.x_rstride + a.T) * 4);
    float sa, sa_inv, sx, sx_inv;
    hg_scale_from_amax(bm_amax_load(a.a_amax), sa, sa_inv);
    hg_scale_from_amax(bm_amax_load(a.x_amax), sx, sx_inv);

    // The pieces of this thread: lane -> (row64 = tid >> 2, quarter pq = tid & 3 of the 16 samples).  Piece
    // i < MW: A row m0 + 64 i + row64; piece MW + j: X row c0 + row64 read at tap shift (j - 1) * dil (KS = 3) or
    // X row c0 + 64 j + row64 (KS = 1).  Byte offsets inside a segment, all >= 0; a row past M / Cn gets
    // 0x7f000000: past the range of every descriptor (segments span < 0x7f000000 bytes, checked by the host) and
    // small enough that adding a chunk offset cannot wrap (reads 0).
    const int row64 = tid >> 2, pq = tid & 3;
    int offa[MW], offx[NS];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = m0 + 64 * i + row64;
        offa[i] = m < a.M ? (m * (int)a.a_rstride + 4 * pq) * 4 : 0x7f000000;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int c = c0 + (KS == 3 ? 0 : 64 * j) + row64;
        offx[j] = c < a.Cn ? (c * (int)a.x_rstride + 4 * pq) * 4 : 0x7f000000;
    }
    // LDS byte address inside a plane: 16-byte slot of the 8-sample half (pq >> 1), 8-byte half (pq & 1)
    const int ldsa = ((pq >> 1) * AQ + row64) * 16 + (pq & 1) * 8;                 // piece i: + i * 64 * 16
    const int ldsx = (ASLOTS + (pq >> 1) * HG_XQ + row64) * 16 + (pq & 1) * 8;     // slot j: + j * 2 * HG_XQ * 16

    f32x16 acc[MW][NS];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[NP], rb[NP];                              // two staging register sets (consumed / in flight)

    int ld_q = 0;
    int ld_s = (int)(q_begin / cps);
    int ld_c = (int)(q_begin - (long)ld_s * cps);
    // segment descriptors, rebuilt only when the cursor enters a new segment
    __amdgpu_buffer_rsrc_t qa = hg_rsrc(a.a + (long)ld_s * a.a_sstride, a_bytes);
    __amdgpu_buffer_rsrc_t qx = hg_rsrc(a.x + (long)ld_s * a.x_sstride, x_bytes);

    // fetches chunk (ld_s, ld_c) into register set R_ and advances the cursor.  Interior chunks: one dwordx4 per
    // piece.  Edge chunks (first / last of a segment, wave-uniform): one dword per sample, samples whose time
    // index is outside [0, T) get an out-of-range offset and read as 0.
#define HG_LOAD(R_)                                                                               \
    {                                                                                             \
        const int t0 = ld_c * HG_K;                                                               \
        if (!(t0 - halo < 0 || t0 + HG_K + halo > a.T)) {                                         \
            const int s0 = __builtin_amdgcn_readfirstlane(t0 * 4);                                \
            _Pragma("unroll") for (int i = 0; i < MW; ++i) R_[i] = hg_ld128(qa, offa[i], s0);     \
            _Pragma("unroll") for (int j = 0; j < NS; ++j)        /* t0 + shift >= 0 in interior chunks */ \
                R_[MW + j] = hg_ld128(qx, offx[j],                                                \
                                      KS == 3 ? __builtin_amdgcn_readfirstlane((t0 + (j - 1) * a.dil) * 4) : s0); \
        } else {                                                                                  \
            unsigned e_[NP][4];       /* all loads first, the register sets are assembled afterwards */ \
            _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                      \
                const __amdgpu_buffer_rsrc_t rs = i < MW ? qa : qx;                               \
                const int shift = (i < MW || KS != 3) ? 0 : (i - MW - 1) * a.dil;                 \
                const int tt0 = t0 + 4 * pq + shift;                                              \
                const int o = (i < MW ? offa[i < MW ? i : 0] : offx[i < MW ? 0 : i - MW]) + (t0 + shift) * 4; \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
                    const int tt = tt0 + r;                                                       \
                    const int oo = (tt >= 0 && tt < a.T) ? o + r * 4 : 0x7ffffff0;                \
                    e_[i][r] = hg_ld32(rs, oo);                                                   \
                }                                                                                 \
            }                                                                                     \
            _Pragma("unroll") for (int i = 0; i < NP; ++i)                                        \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) R_[i][r] = e_[i][r];                \
        }                                                                                         \
        if (++ld_q < nst) {                            /* else: stays on the last chunk */        \
            if (++ld_c == cps) {                                                                  \
                ld_c = 0; ++ld_s;                                                                 \
                qa = hg_rsrc(a.a + (long)ld_s * a.a_sstride, a_bytes);                            \
                qx = hg_rsrc(a.x + (long)ld_s * a.x_sstride, x_bytes);                            \
            }                                                                                     \
        }                                                                                         \
    }
    // splits piece I_ of register set R_ into buffer BUF_ (two 8-byte LDS writes, immediate offsets)
#define HG_STORE(R_, I_, BUF_)                                                                    \
    {                                                                                             \
        char* dst_ = reinterpret_cast<char*>(lds + (BUF_) * BUF) +                                \
                     ((I_) < MW ? ldsa + (I_) * 64 * 16 : ldsx + ((I_) - MW) * 2 * HG_XQ * 16);   \
        float f_[4];                                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) f_[r] = __uint_as_float(R_[I_][r]);         \
        hg_split_store4(f_, (I_) < MW ? sa : sx, dst_, PLANE * 16);                               \
    }
#define HG_STORE_ALL(R_, BUF_)                                                                    \
    {                                                                                             \
        HG_STORE(R_, 0, BUF_) HG_STORE(R_, 1, BUF_) HG_STORE(R_, 2, BUF_) HG_STORE(R_, 3, BUF_)   \
        HG_STORE(R_, 4, BUF_) HG_STORE(R_, 5, BUF_)                                               \
        if constexpr (NP == 8) { HG_STORE(R_, 6, BUF_) HG_STORE(R_, 7, BUF_) }                    \
    }
#define HG_TERM(PA_, PB_)                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        _Pragma("unroll") for (int j = 0; j < NS; ++j)                                            \
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA_][mt], bf[PB_][j], acc[mt][j], 0, 0, 0);
    // Operand fragments live across stages: [plane 0 = hi, 1 = lo].
    f16x8 af[2][MW], bf[2][NS];
    // fragments a stage needs first (A.lo, B.hi), read from buffer BUF_ as soon as it is complete
#define HG_FRAGS_EARLY(BUF_)                                                                      \
    {                                                                                             \
        const u32x4* pb = lds + (BUF_) * BUF;                                                     \
        const u32x4* ab = pb + h * AQ + wm * (MW * 32) + nl;                                      \
        const u32x4* xb = pb + ASLOTS + h * HG_XQ + wc * 32 + nl;                                 \
        _Pragma("unroll") for (int j = 0; j < NS; ++j) bf[0][j] = __builtin_bit_cast(f16x8, xb[j * 2 * HG_XQ]); \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[1][mt] = __builtin_bit_cast(f16x8, ab[PLANE + mt * 32]); \
    }
    // One stage k on buffer BUF_; P_ holds chunk k + 1 (fetched a stage ago), Q_ receives chunk k + 2.  Every
    // stage does all of it, the last ones on clamped (repeated) chunks whose results are never read.
#define HG_STAGE(BUF_, NBUF_, P_, Q_)                                                             \
    {                                                                                             \
        hg_touch<NP>(P_);             /* the one wait of the stage */                             \
        HG_LOAD(Q_)                   /* its interior / edge branch ends the scheduling region */ \
        const u32x4* pb = lds + (BUF_) * BUF;                                                     \
        const u32x4* ab = pb + h * AQ + wm * (MW * 32) + nl;                                      \
        const u32x4* xb = pb + ASLOTS + h * HG_XQ + wc * 32 + nl;                                 \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[0][mt] = __builtin_bit_cast(f16x8, ab[mt * 32]); \
        _Pragma("unroll") for (int j = 0; j < NS; ++j) bf[1][j] = __builtin_bit_cast(f16x8, xb[PLANE + j * 2 * HG_XQ]); \
        HG_STORE_ALL(P_, NBUF_)                                                                   \
        HG_TERM(1, 0)                                                                             \
        HG_TERM(0, 0)                                                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 2 * MW * NS; ++g_) {                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                    \
            if (g_ & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        \
        }                                                                                         \
        hg_barrier<MW, NS>(af[0], bf[0]);                                                         \
        HG_FRAGS_EARLY(NBUF_)                                                                     \
        HG_TERM(0, 1)                                                                             \
    }

    if (nst > 0) {
        HG_LOAD(ra)
        HG_LOAD(rb)
        HG_STORE_ALL(ra, 0)
        __syncthreads();
        HG_FRAGS_EARLY(0)
        // stages in pairs (the two register sets and the two LDS buffers swap roles every stage), an odd last
        // stage after the loop: no conditional inside the loop body
        for (int k = 0; k + 1 < nst; k += 2) {
            HG_STAGE(0, 1, rb, ra)
            HG_STAGE(1, 0, ra, rb)
        }
        if (nst & 1) {                 // last stage: nothing left to fetch or split
            const u32x4* pb = lds;
            const u32x4* ab = pb + h * AQ + wm * (MW * 32) + nl;
            const u32x4* xb = pb + ASLOTS + h * HG_XQ + wc * 32 + nl;
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) af[0][mt] = __builtin_bit_cast(f16x8, ab[mt * 32]);
#pragma unroll
            for (int j = 0; j < NS; ++j) bf[1][j] = __builtin_bit_cast(f16x8, xb[PLANE + j * 2 * HG_XQ]);
            HG_TERM(1, 0)
            HG_TERM(0, 0)
            HG_TERM(0, 1)
        }
    }
#undef HG_LOAD
#undef HG_FRAGS_EARLY
#undef HG_STORE
#undef HG_STORE_ALL
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
                    for (int j = 0; j < NS; ++j) {
                        const int c = c0 + 64 * j + wc * 32 + nl;
                        if (c < a.Cn) dst[(long)m * N + c] = acc[mt][j][r] * f;
                    }
                }
            }
        }
    }
}

// tile family of a shape: 0 = not covered, 1 = 320 x 192 (<KS,5,3>), 2 = 256 x 128 (<1,4,2>)
static int hg_family(int S, int G, int M, int Cn, int T, int KS, int dil, bool ordered) {
    static int wide = -1;
    if (wide < 0) {
        const char* e = getenv("BM_H2_WIDE");          // BM_H2_WIDE=0: A/B runs against the 3 x bf16 kernels
        wide = !(e && e[0] == '0');
    }
    if (!wide || G != 1 || ordered || (KS != 3 && KS != 1)) return 0;
    if (dil < 1 || dil > 32 || T < 2 * HG_K) return 0;
    if ((long)S * ((T + HG_K - 1) / HG_K) < 64) return 0;
    // padded rows / columns are wasted MFMA work: at most 25 % (3 taps) / 50 % (1x1 layers, small in absolute terms)
    const long work = (long)M * Cn;
    const long pad53 = (long)cdiv(M, 320) * 320 * cdiv(Cn, KS == 3 ? 64 : 192) * (KS == 3 ? 64 : 192);
    if (KS == 3) return pad53 * 4 <= work * 5 ? 1 : 0;
    const long pad42 = (long)cdiv(M, 256) * 256 * cdiv(Cn, 128) * 128;
    if (pad42 < pad53 && pad42 <= work * 2) return 2;
    return pad53 <= work * 2 ? 1 : 0;
}

extern "C" int bm_gemm_nt_h2_covers(int M, int Cn, int KS, int S, int T, int G, int dil, int ordered) {
    return hg_family(S, G, M, Cn, T, KS, dil, ordered != 0) ? 1 : 0;
}

// one workgroup per CU per round (256 CUs), >= 16 stages per workgroup
extern "C" int bm_gemm_nt_h2_suggest_splits(int M, int Cn, int KS, int S, int T) {
    const int fam = hg_family(S, 1, M, Cn, T, KS, 1, false);
    const int tiles = fam == 2 ? cdiv(M, 256) * cdiv(Cn, 128) : cdiv(M, 320) * cdiv(Cn, KS == 3 ? 64 : 192);
    const long chunks = (long)S * ((T + HG_K - 1) / HG_K);
    long want = 256 / tiles;
    if (want > chunks / 16) want = chunks / 16;
    if (want < 1) want = 1;
    return (int)want;
}

template <int KS, int MW, int NS>
static int launch_gemm_nt_h2w(GemmNTArgsH g, hipStream_t stream) {
    constexpr int PLANE = 2 * (64 * MW + 8) + NS * 2 * HG_XQ;
    const size_t lds = (size_t)2 * 2 * PLANE * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_h2w_kernel<KS, MW, NS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "gemm_nt_h2w: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    g.tiles_m = cdiv(g.M, 64 * MW);
    g.tiles_c = cdiv(g.Cn, KS == 3 ? 64 : 64 * NS);
    const long nblocks = (long)g.tiles_m * g.tiles_c * g.nsplit;
    hipLaunchKernelGGL((gemm_nt_h2w_kernel<KS, MW, NS>), dim3((unsigned)nblocks), dim3(256), lds, stream, g);
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
    const int fam = hg_family(S, 1, M, Cn, T, KS, dil, false);
    BM_REQUIRE(fam != 0, "gemm_nt_h2: shape not covered (M=%d Cn=%d T=%d KS=%d dil=%d)", M, Cn, T, KS, dil);
    BM_REQUIRE(((long)(M - 1) * a_rstride + T) * 4 < 0x7f000000L && ((long)(Cn - 1) * x_rstride + T) * 4 < 0x7f000000L,
               "gemm_nt_h2: a segment spans 2 GB or more");
    GemmNTArgsH g;
    g.a = a; g.a_sstride = a_sstride; g.a_rstride = a_rstride; g.a_amax = a_amax;
    g.x = x; g.x_sstride = x_sstride; g.x_rstride = x_rstride; g.x_amax = x_amax;
    g.part = part; g.S = S; g.M = M; g.Cn = Cn; g.T = T; g.dil = dil; g.nsplit = nsplit;
    hipStream_t s = (hipStream_t)stream;
    if (KS == 3) return launch_gemm_nt_h2w<3, 5, 3>(g, s);
    return fam == 2 ? launch_gemm_nt_h2w<1, 4, 2>(g, s) : launch_gemm_nt_h2w<1, 5, 3>(g, s);
}
