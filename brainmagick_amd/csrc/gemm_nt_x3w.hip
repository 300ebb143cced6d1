// Wide-tile variant of the fp32-accurate ("f32x3") time-contraction GEMM of gemm_nt_x3.hip for the
// weight gradients of the conv stack (3 taps, one group; M a multiple of 320, Cn padded to a multiple of 64
// by the bounds check when that wastes at most a quarter of the work):
//
//   part[split][m][c*3 + j] = sum_{s, t in split} A[s][m][t] * X[s][c][t + (j - 1) * dil]
//
// ONE workgroup of four wavefronts per CU (one per SIMD); a wavefront owns 160 rows x (32 channels x 3
// taps) as 5 x 3 MFMA accumulators (240 registers), the workgroup 320 x (64 x 3).  With 1 x 1 x 3 blocks
// per wavefront (gemm_nt_x3.hip) the exact 3-way bf16 split of the operands costs ~9 VALU instructions
// per MFMA and the kernel is VALU-bound; here it is ~2 per MFMA and is interleaved with the MFMAs of
// the same wavefront.  Stage = 16 samples: 4 items of 8 samples per thread (A rows 0..319, then the three
// tap-shifted copies of the 64 X rows), fetched three stages ahead with bounds-checked dwordx4 buffer loads
// (dword alignment suffices, so every tap shift takes the wide path), split and written into the other
// LDS buffer while the 90 MFMAs of the current stage run.
// LDS (16-byte slots = 8 samples of one plane), three stage buffers of three planes each:
//   A [2 halves of the 16 samples][320 + 8 rows], X [3 taps][2 halves][64 + 8 rows]   (153 KB in all).
// Rows are contiguous inside a half (the row-per-lane ds_read_b128 of an MFMA operand reads 512 contiguous
// bytes per half-wavefront); the 8 pad rows put the two halves 32 banks apart for the staging writes
// (adjacent lanes = the two halves of one row).  rocprofv3: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33,
// LDS ~19 % busy - not what bounds the kernel.
#include <cstdlib>
#include "bm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GW_BM 320
#define GW_BC 64
#define GW_K 16
#define GW_AQ (GW_BM + 8)                     // slots of one 8-sample half of the A tile (8 pad rows)
#define GW_XQ (GW_BC + 8)                     // ... of one tap of the X tile
#define GW_ASLOTS (2 * GW_AQ)                 // 16-byte slots of one plane of the A tile
#define GW_XSLOTS (3 * 2 * GW_XQ)             // ... of the X tile (3 taps)
#define GW_BUF (3 * (GW_ASLOTS + GW_XSLOTS))  // slots of one stage buffer (3 planes)

struct GemmNTArgsW {
    const float* a; long a_sstride; long a_rstride;
    const float* x; long x_sstride; long x_rstride;
    float* part;
    int S, M, Cn, T, dil, nsplit;
    int tiles_m, tiles_c;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (base, stride 0, num_records = bytes, 32-bit data format) in four SGPRs
__device__ __forceinline__ i32x4 gw_rsrc(const float* p, int bytes) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
    d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(u >> 32) & 0xffffu));
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000;
    return d;
}

// The staging loads are issued through inline asm so that the compiler's waitcnt pass does not see them: it
// would drain the queue (vmcnt(0)) at the first use of the older register set and so stall every stage on
// the loads it has just issued.  The matching counted waits are GW_WAIT below; until then the destination
// registers must not be read (they are only read by the split that follows the wait).
// interior chunks: per-thread byte offset (constant for the whole kernel) + wave-uniform byte offset of the
// chunk in an SGPR + 0 / 16 as immediate: no address arithmetic per load
__device__ __forceinline__ u32x4 gw_ld128(i32x4 rs, int voff, int soff) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ unsigned gw_ld32(i32x4 rs, int voff) {
    unsigned v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}

// exact 3-way split of 4 fp32 values into bf16 planes; one 8-byte store per plane tile
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gw_split_store4(const float (&f)[4], char* dst, int plane_stride_bytes) {
    bf16x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 a = (__bf16)f[i];
        const float r1 = f[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[i] = a; m[i] = b; l[i] = (__bf16)(r1 - (float)b);
    }
    *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2*>(dst + plane_stride_bytes) = __builtin_bit_cast(u32x2, m);
    *reinterpret_cast<u32x2*>(dst + 2 * plane_stride_bytes) = __builtin_bit_cast(u32x2, l);
}

__global__ __launch_bounds__(256, 1) void gemm_nt_x3w_kernel(GemmNTArgsW a) {
    constexpr int MW = 5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);      // [2 buffers][3 planes][A slots | X slots]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wc = wave & 1;
    const int nl = lane & 31, h = lane >> 5;

    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % a.tiles_m; id /= a.tiles_m;
    const int tc = id % a.tiles_c;
    const int split = id / a.tiles_c;
    const int m0 = tm * GW_BM, c0 = tc * GW_BC;

    const int cps = (a.T + GW_K - 1) / GW_K;
    const long nchunks = (long)a.S * cps;
    const long q_begin = nchunks * split / a.nsplit;
    const long q_end = nchunks * (split + 1) / a.nsplit;
    const int nst = (int)(q_end - q_begin);
    const int halo = a.dil;
    const int a_bytes = (int)(((long)(a.M - 1) * a.a_rstride + a.T) * 4);
    const int x_bytes = (int)(((long)(a.Cn - 1) * a.x_rstride + a.T) * 4);

    // The eight 4-sample pieces of this thread: lane -> (row64 = tid >> 2, quarter p = tid & 3 of the 16
    // samples).  Piece i < 5: A row m0 + 64 i + row64; piece 5 + j: X row c0 + row64 read at tap shift
    // (j - 1) * dil.  Which operand / tap a piece belongs to is a compile-time property of i.
    const int row64 = tid >> 2, pq = tid & 3;
    const int offa = ((m0 + row64) * (int)a.a_rstride + 4 * pq) * 4;   // byte offsets inside a segment (>= 0)
    const int offx = ((c0 + row64) * (int)a.x_rstride + 4 * pq) * 4;
    const int a_step = 64 * (int)a.a_rstride * 4;                      // piece i of A: + i * a_step (scalar)
    // LDS byte address inside a plane: 16-byte slot of the 8-sample half (pq >> 1), 8-byte half (pq & 1)
    const int ldsa = ((pq >> 1) * GW_AQ + row64) * 16 + (pq & 1) * 8;                 // piece i: + i * 64 * 16
    const int ldsx = (GW_ASLOTS + (pq >> 1) * GW_XQ + row64) * 16 + (pq & 1) * 8;     // tap j: + j * 2 * GW_XQ * 16

    f32x16 acc[MW][3];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[8], rb[8], rc[8];                         // three staging register sets (chunks k+1 .. k+3)

#ifdef GW_PROFILE
    long long pt[5] = {0, 0, 0, 0, 0};
    long long plast = clock64();
    const long long pstart = plast;
#define GW_PSTAMP(I_) { const long long now_ = clock64(); pt[I_] += now_ - plast; plast = now_; }
#else
#define GW_PSTAMP(I_)
#endif
    int ld_q = 0;
    int ld_s = (int)(q_begin / cps);
    int ld_c = (int)(q_begin - (long)ld_s * cps);
    // segment descriptors, rebuilt only when the cursor enters a new segment
    i32x4 qa = gw_rsrc(a.a + (long)ld_s * a.a_sstride, a_bytes);
    i32x4 qx = gw_rsrc(a.x + (long)ld_s * a.x_sstride, x_bytes);

    // loads chunk (ld_s, ld_c) into register set R_ and advances the cursor.  Interior chunks: two dwordx4
    // per item.  Edge chunks (first / last of a segment, wave-uniform): one dword per sample, samples whose
    // time index is outside [0, T) get an out-of-range offset and read as 0.
#define GW_LOAD(R_)                                                                               \
    {                                                                                             \
        const int t0 = ld_c * GW_K;                                                               \
        if (!(t0 - halo < 0 || t0 + GW_K + halo > a.T)) {                                         \
            _Pragma("unroll") for (int i = 0; i < 5; ++i)                                         \
                R_[i] = gw_ld128(qa, offa, __builtin_amdgcn_readfirstlane(t0 * 4 + i * a_step));  \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)         /* t0 + shift >= 0 in interior chunks */ \
                R_[5 + j] = gw_ld128(qx, offx, __builtin_amdgcn_readfirstlane((t0 + (j - 1) * a.dil) * 4)); \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                       \
                const i32x4 rs = i < 5 ? qa : qx;                                                 \
                const int shift = i < 5 ? 0 : (i - 6) * a.dil;                                    \
                const int tt0 = t0 + 4 * pq + shift;                                              \
                const int o = (i < 5 ? offa + i * a_step : offx) + (t0 + shift) * 4;              \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
                    const int tt = tt0 + r;                                                       \
                    const int oo = (tt >= 0 && tt < a.T) ? o + r * 4 : 0x7ffffff0;                \
                    R_[i][r] = gw_ld32(rs, oo);                                                   \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
        if (++ld_q < nst) {                            /* else: stays on the last chunk */        \
            if (++ld_c == cps) {                                                                  \
                ld_c = 0; ++ld_s;                                                                 \
                qa = gw_rsrc(a.a + (long)ld_s * a.a_sstride, a_bytes);                            \
                qx = gw_rsrc(a.x + (long)ld_s * a.x_sstride, x_bytes);                            \
            }                                                                                     \
        }                                                                                         \
    }
    // counted wait for register set R_: at most N_ younger load instructions may stay in flight.  (After an
    // edge chunk 32 instead of 8 younger loads are outstanding: vmcnt(8) then over-waits, which is safe.)
#define GW_WAIT(R_, N_)                                                                           \
    asm volatile("s_waitcnt vmcnt(" #N_ ")"                                                       \
                 : "+v"(R_[0]), "+v"(R_[1]), "+v"(R_[2]), "+v"(R_[3]), "+v"(R_[4]), "+v"(R_[5]),  \
                   "+v"(R_[6]), "+v"(R_[7])::"memory");
    // splits piece I_ of register set R_ into buffer BUF_ (three 8-byte LDS writes, immediate offsets)
#define GW_STORE(R_, I_, BUF_)                                                                    \
    {                                                                                             \
        char* dst_ = reinterpret_cast<char*>(lds + (BUF_) * GW_BUF) +                             \
                     ((I_) < 5 ? ldsa + (I_) * 64 * 16 : ldsx + ((I_) - 5) * 2 * GW_XQ * 16);     \
        float f_[4];                                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) f_[r] = __uint_as_float(R_[I_][r]);         \
        gw_split_store4(f_, dst_, (GW_ASLOTS + GW_XSLOTS) * 16);                                  \
    }
#define GW_TERM(PA_, PB_)                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                             \
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA_][mt], bf[PB_][j], acc[mt][j], 0, 0, 0);
    // Operand fragments live across stages: [plane 0 = hi, 1 = mid, 2 = lo].
    bf16x8 af[3][MW], bf[3][3];
    // fragments a stage needs first (B.hi, A.lo, B.lo), read from buffer BUF_ as soon as it is complete
#define GW_FRAGS_EARLY(BUF_)                                                                      \
    {                                                                                             \
        const u32x4* pb = lds + (BUF_) * GW_BUF;                                                  \
        const u32x4* ab = pb + h * GW_AQ + wm * (MW * 32) + nl;                                   \
        const u32x4* xb = pb + GW_ASLOTS + h * GW_XQ + wc * 32 + nl;                              \
        constexpr int PS = GW_ASLOTS + GW_XSLOTS;                                                 \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) bf[0][j] = __builtin_bit_cast(bf16x8, xb[0 * PS + j * 2 * GW_XQ]); \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[2][mt] = __builtin_bit_cast(bf16x8, ab[2 * PS + mt * 32]); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) bf[2][j] = __builtin_bit_cast(bf16x8, xb[2 * PS + j * 2 * GW_XQ]); \
    }
    // One stage k.  On entry B.hi, A.lo, B.lo of buffer BUF_ are already in registers (read right after the
    // barrier of stage k - 1, under its last 30 MFMAs).  Pieces 0-5 of chunk k + 1 (register set ST_) are split
    // into buffer NBUF_ between the first 60 MFMAs (lo*hi, hi*lo, mid*hi, hi*hi); then the barrier, the early
    // fragments of stage k + 1, pieces 6-7 of chunk k + 2 (set S2_) into buffer N2BUF_, the fetch of chunk k + 3
    // into register set LD_ (which held chunk k), and the last 30 MFMAs (mid*mid, hi*mid).  Every stage does all of it, the last ones on clamped (repeated) chunks whose
    // results are never read.  The order of the six partial products inside a stage is immaterial for the
    // rounding: each is added to the running fp32 sum over all earlier stages.
#define GW_STAGE(BUF_, NBUF_, N2BUF_, LD_, ST_, S2_)                                              \
    {                                                                                             \
        const u32x4* pb = lds + (BUF_) * GW_BUF;                                                  \
        const u32x4* ab = pb + h * GW_AQ + wm * (MW * 32) + nl;                                   \
        const u32x4* xb = pb + GW_ASLOTS + h * GW_XQ + wc * 32 + nl;                              \
        constexpr int PS = GW_ASLOTS + GW_XSLOTS;                                                 \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[0][mt] = __builtin_bit_cast(bf16x8, ab[0 * PS + mt * 32]); \
        GW_WAIT(ST_, 8)               /* chunk k + 1 (landed a stage ago); chunk k + 2 may stay in flight */ \
        GW_STORE(ST_, 0, NBUF_)                                                                   \
        GW_STORE(ST_, 1, NBUF_)                                                                   \
        GW_TERM(2, 0)                                                                             \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) bf[1][j] = __builtin_bit_cast(bf16x8, xb[1 * PS + j * 2 * GW_XQ]); \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[1][mt] = __builtin_bit_cast(bf16x8, ab[1 * PS + mt * 32]); \
        GW_STORE(ST_, 2, NBUF_)                                                                   \
        GW_STORE(ST_, 3, NBUF_)                                                                   \
        GW_TERM(0, 2)                                                                             \
        GW_STORE(ST_, 4, NBUF_)                                                                   \
        GW_STORE(ST_, 5, NBUF_)                                                                   \
        GW_TERM(1, 0)                                                                             \
        GW_TERM(0, 0)                                                                             \
        /* spread the split arithmetic between the MFMAs: one MFMA, then up to three VALU ops (a VALU */ \
        /* op holds the issue port for 8 cycles, an MFMA for 8 of its 32: scripts/micro/mfma_valu_mix.hip) */ \
        _Pragma("unroll") for (int g_ = 0; g_ < 60; ++g_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                    \
        }                                                                                         \
        GW_PSTAMP(0)                                                                              \
        __syncthreads();                                                                          \
        GW_PSTAMP(1)                                                                              \
        GW_FRAGS_EARLY(NBUF_)                                                                     \
        GW_PSTAMP(2)                                                                              \
        GW_WAIT(S2_, 0)               /* chunk k + 2, requested one stage ago */                  \
        GW_PSTAMP(3)                                                                              \
        GW_STORE(S2_, 6, N2BUF_)      /* its last two pieces go under the last 30 MFMAs */        \
        GW_STORE(S2_, 7, N2BUF_)                                                                  \
        GW_LOAD(LD_)                                                                              \
        GW_TERM(1, 1)                                                                             \
        GW_TERM(0, 1)                                                                             \
        _Pragma("unroll") for (int g_ = 0; g_ < 30; ++g_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                    \
        }                                                                                         \
        GW_PSTAMP(4)                                                                              \
    }

    if (nst > 0) {
        GW_LOAD(ra)
        GW_LOAD(rb)
        GW_LOAD(rc)
        GW_WAIT(ra, 0)
        GW_WAIT(rb, 0)
        GW_WAIT(rc, 0)
        GW_STORE(ra, 0, 0) GW_STORE(ra, 1, 0) GW_STORE(ra, 2, 0) GW_STORE(ra, 3, 0)
        GW_STORE(ra, 4, 0) GW_STORE(ra, 5, 0) GW_STORE(ra, 6, 0) GW_STORE(ra, 7, 0)
        GW_STORE(rb, 6, 1) GW_STORE(rb, 7, 1)
        __syncthreads();
        GW_FRAGS_EARLY(0)
        // stage k reads buffer k % 3; register set k % 3 held chunk k and now receives chunk k + 3.  Pieces 0-5
        // of chunk k + 1 are split before the barrier of stage k, its pieces 6-7 were split after the barrier
        // of stage k - 1 (into a buffer nobody reads before barrier k).
        for (int k = 0; k < nst; k += 3) {
            GW_STAGE(0, 1, 2, ra, rb, rc)
            if (k + 1 < nst) GW_STAGE(1, 2, 0, rb, rc, ra)
            if (k + 2 < nst) GW_STAGE(2, 0, 1, rc, ra, rb)
        }
        // The compiler does not know about the asm-issued fetches: the (unused) ones of the last stages are
        // still in flight and would land in registers it hands to the epilogue.  Drain them here.
        GW_WAIT(ra, 0)
        GW_WAIT(rb, 0)
        GW_WAIT(rc, 0)
    }
#undef GW_LOAD
#undef GW_FRAGS_EARLY
#undef GW_WAIT
#undef GW_STORE
#undef GW_TERM
#undef GW_STAGE

#ifdef GW_PROFILE
    const long long ptotal = clock64() - pstart;
#endif
    // partial tile out: part[split][m][c * 3 + j]
    const long N = (long)a.Cn * 3;
    float* dst = a.part + (long)split * a.M * N;
    const int c = c0 + wc * 32 + nl;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (MW * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < a.M && c < a.Cn) {
                float* p = dst + (long)m * N + (long)c * 3;
                p[0] = acc[mt][0][r];
                p[1] = acc[mt][1][r];
                p[2] = acc[mt][2][r];
            }
        }
    }
#ifdef GW_PROFILE
    __syncthreads();
    if (blockIdx.x == 0 && lane == 0) {     // overwrites the head of the partial buffer: profiling builds only
        float* d = a.part + wave * 8;
        for (int i = 0; i < 5; ++i) d[i] = (float)(pt[i] / nst);
        d[5] = (float)(ptotal / nst);
        d[6] = (float)nst;
    }
#endif
}

static bool gw_covers(int S, int G, int M, int Cn, int T, int KS, int dil, bool ordered) {
    static int wide = -1;
    if (wide < 0) {
        const char* e = getenv("BM_X3_WIDE");          // BM_X3_WIDE=0: A/B runs against gemm_nt_x3_kernel
        wide = !(e && e[0] == '0');
    }
    if (!wide || G != 1 || ordered || KS != 3) return false;
    if (dil < 1 || dil > 32 || T < 2 * GW_K) return false;
    // rows past M / Cn read as zeros (their byte offset alone is past the descriptor's range) and are never
    // written; take the wide kernel while the padded tile grid wastes at most a quarter of the work
    // (the X rows are addressed through the per-lane voffset, which the descriptor range-checks; the A rows
    // of pieces 1..4 are addressed through the scalar soffset, which is NOT range-checked, so A is never
    // padded: M must fill its 320-row tiles)
    if (M % GW_BM != 0) return false;
    const long padded = (long)cdiv(M, GW_BM) * GW_BM * cdiv(Cn, GW_BC) * GW_BC;
    if (padded * 4 > (long)M * Cn * 5) return false;
    return (long)S * ((T + GW_K - 1) / GW_K) >= 64;
}

// Split count for bm_gemm_nt_x3: when the wide kernel covers the shape, one workgroup per CU per round
// (256 CUs); otherwise the generic rule of bm_gemm_nt_suggest_splits.
extern "C" int bm_gemm_nt_suggest_splits(int M, int Cn, int KS, int S, int T, int G);
extern "C" int bm_gemm_nt_x3_suggest_splits(int M, int Cn, int KS, int S, int T, int G, int dil) {
    if (!gw_covers(S, G, M, Cn, T, KS, dil, false)) return bm_gemm_nt_suggest_splits(M, Cn, KS, S, T, G);
    const int tiles = cdiv(M, GW_BM) * cdiv(Cn, GW_BC);
    long chunks = (long)S * ((T + GW_K - 1) / GW_K);
    long want = 256 / tiles;
    if (want < 1) want = 1;
    if (want > chunks / 32) want = chunks / 32;        // >= 32 stages per workgroup
    if (want < 1) want = 1;
    return (int)want;
}

// returns -1 when the shape is not covered (the caller then takes gemm_nt_x3_kernel)
int bm_gemm_nt_x3w_try(const float* a, long a_sstride, long a_rstride, const float* x, long x_sstride,
                       long x_rstride, const int* order, float* part, int S, int G, int M, int Cn, int T,
                       int KS, int dil, int nsplit, hipStream_t stream) {
    if (!gw_covers(S, G, M, Cn, T, KS, dil, order != nullptr)) return -1;
    if ((long)M * a_rstride * 4 >= 0x7f000000L || (long)Cn * x_rstride * 4 >= 0x7f000000L) return -1;
    GemmNTArgsW g;
    g.a = a; g.a_sstride = a_sstride; g.a_rstride = a_rstride;
    g.x = x; g.x_sstride = x_sstride; g.x_rstride = x_rstride;
    g.part = part; g.S = S; g.M = M; g.Cn = Cn; g.T = T; g.dil = dil; g.nsplit = nsplit;
    g.tiles_m = cdiv(M, GW_BM); g.tiles_c = cdiv(Cn, GW_BC);
    const size_t lds = (size_t)3 * GW_BUF * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_x3w_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "gemm_nt_x3w: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long nblocks = (long)g.tiles_m * g.tiles_c * nsplit;
    hipLaunchKernelGGL(gemm_nt_x3w_kernel, dim3((unsigned)nblocks), dim3(256), lds, stream, g);
    return bm_check_launch("gemm_nt_x3w");
}
