// Wide-tile time-contraction GEMM in compute mode "f16x2" (see conv_nn_h2w.hip for the arithmetic): the weight
// gradients of the conv stack and of the 1x1 layers, and the ClipLoss score contraction,
//
//   KS = 3:  part[split][m][c*3 + j] = sum_{s, t in split} A[s][m][t] * X[s][c][t + (j - 1) * dil]
//   KS = 1:  part[split][m][c]       = sum_{s, t in split} A[s][m][t] * X[s][c][t]
//
// both operands activations: each is scaled by a power of two taken from its tensor maximum (a_amax / x_amax,
// device amax slots), split into two f16 planes while it is staged, and the partial tile is multiplied by the
// exact inverse scales on the way out.  Three MFMAs per 32x32x16 block (lo*hi, hi*hi, hi*lo).
//
// ONE workgroup of four wavefronts per CU (one per SIMD); a wavefront owns (32 MW) rows x (32 columns x NS
// "slots") as MW x NS MFMA accumulators; the slots are the three taps of 64 X rows (KS = 3, NS = 3) or NS blocks
// of 64 X rows (KS = 1).  Tiles: <3,5,3> and <1,5,3> 320 x 192 (weight gradients), <1,4,2> 256 x 128, and
// <1,4,4,TS> 256 x 256 with a TRANSPOSED partial tile (ClipLoss scores, bm_clip_scores_h2: A = candidates, X =
// estimates, part[split][estimate][candidate] -- a lane's four consecutive accumulator rows are four consecutive
// candidates of one estimate, stored as one 16-byte write instead of four scattered dwords).
//
// Stage = 32 samples = two MFMA k-steps.  A stage's operands are fetched as FULL 128-byte lines: eight lanes per
// row (4 samples = 16 bytes each), 32 rows per wave-load pass, 2 MW + 2 NS passes per thread.  (With 16-sample
// stages a wave-load touches 16 half-lines and the per-CU load path, not the matrix pipe, bounds the kernel:
// measured 8 bytes / clock / CU, 40 % matrix-pipe busy.)  ONE staging register set; pipeline of stage k (MFMAs on
// LDS buffer k & 1):
//     wait for chunk k + 1 (fetched half a stage ago)  ->  split it into LDS buffer (k + 1) & 1 between the MFMAs
//     of k-step 0  ->  fetch chunk k + 2 into the same registers  ->  k-step 1: lo*hi, hi*hi  ->  barrier  ->
//     early fragments of stage k + 1 under the last term (hi*lo).
// Operand fragments roll through one register set: a term's fragments are replaced by those of the next k-step
// as soon as its MFMAs are issued, so every LDS read is covered by the 15 MFMAs of the term in front of its use.
// The loads are compiler-visible raw buffer loads: gemm_nt_x3w.hip hides its loads in inline asm with hand-counted
// waits, which is only sound while the register allocator never copies a register whose load is still in
// flight -- it did here (v_mov of a staging set ahead of its wait).  hipcc's own waits are conservative around the
// interior / edge branch of the fetch; with one wait per stage, placed where every outstanding load is at least
// half a stage old, conservative costs nothing.
// Rows past M / Cn are addressed through the per-lane offset, which the buffer descriptor range-checks: they
// read as zeros and are never written, so M and Cn may be padded.
// LDS (16-byte slots = 8 samples of one plane), two stage buffers of two planes each:
//   A [4 groups of 8 samples][64 MW + 2 rows], X [NS slots][4 groups][64 + 2 rows]   (row counts = 2 mod 8: the
//   16 lanes of a ds_write_b64 group -- 2 rows x 8 pieces -- land on 16 different bank pairs).
#include <cstdlib>
#include <utility>
#include <cstring>
#include "bm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define HG_K 32                               // samples per stage
#define HG_XQ (64 + 2)                        // slots of one 8-sample group of one X slot (2 pad rows)

struct GemmNTArgsH {
    const float* a; long a_sstride; long a_rstride;
    const float* x; long x_sstride; long x_rstride;
    const float* a_amax; const float* x_amax;
    const float* a_row_amax;      // RS kernels: [M] per-row max |a| (the rows of dW factor out of the contraction)
    float* part;
    int S, M, Cn, T, dil, nsplit;
    int tiles_m, tiles_c;
    // grouped contraction (per-(layout, subject) weight gradient of the composed front end): group g contracts the
    // segments order[seg[g] .. seg[g + 1]) (order nullable = identity); part[g][split][M][N].  Null seg: one group
    // of all S segments.  Not with the flat walk.
    const int* order; const int* seg;
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

// two fp32 values -> scaled f16 pairs: hi = f16(x * s), lo = f16(x * s - hi) (the product is exact, s is a power
// of two; the difference is exact in fp32), written straight into the halves of the packed results: 4 VALU
__device__ __forceinline__ void hg_split_pair(float x0, float x1, float s0, float s1, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %5, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(x0), "v"(x1), "v"(s0), "v"(s1));
}

// the two halves of hg_split_pair as separate statements (2 VALU each): the main loop puts them behind different MFMAs
__device__ __forceinline__ void hg_split_hi(float x0, float x1, float s0, float s1, unsigned& hi) {
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %4, 0" : "=&v"(hi) : "v"(x0), "v"(x1), "v"(s0), "v"(s1));
}
__device__ __forceinline__ void hg_split_lo(float x0, float x1, float s0, float s1, unsigned hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %5, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo) : "v"(hi), "v"(x0), "v"(x1), "v"(s0), "v"(s1));
}

// lgkmcnt(0) + workgroup barrier; names A.hi / B.hi as operands so that the register-only MFMAs that read them
// stay on their side of the barrier (see conv_nn_h2w.hip)
template <int MW, int NS>
__device__ __forceinline__ void hg_barrier(f16x8 (&ah)[MW], f16x8 (&bh)[NS]) {
    if constexpr (MW == 5 && NS == 3)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2])
                     :: "memory");
    else if constexpr (MW == 4 && NS == 4)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3])
                     :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(bh[0]), "+v"(bh[1])
                     :: "memory");
}

// "every register of the staging set is needed now": the compiler places its wait for the whole set here
template <int NP>
__device__ __forceinline__ void hg_touch(u32x4 (&r)[NP]) {
    if constexpr (NP == 16)
        asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]),
                     "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]));
    else
        asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]),
                     "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]));
}

template <int... I, class F>
__device__ __forceinline__ void hg_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void hg_static_for(F&& f) {
    hg_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#ifdef HG_TRACE
// cycle trace of the stage pipeline (diagnostic builds only, scripts/build_trace_lib.sh): per workgroup and
// wavefront, cycles summed over the stages of seven segments of the loop body
__device__ long long hg_trace_buf[64 * 4 * 8];
#define HG_T(I_) { __builtin_amdgcn_sched_barrier(0); tr[I_] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int bm_debug_trace_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hg_trace_buf), sizeof(long long) * 64 * 4 * 8);
}
#else
#define HG_T(I_)
#endif

// RS ("row scales"): every row of A carries its own power-of-two scale, taken from a per-row maximum that the
// producer of A published (act_bn_bwd / glu_bwd: one gradient channel = one row of dW).  A channel far below its
// tensor's maximum then keeps its 22 bits.  Needs T % 4 == 0: a 4-sample piece is then entirely
// inside or outside a row, pieces outside are fetched at an out-of-range offset (zeros), and A needs no per-value
// edge scales.
// TS ("transposed store"): part[split][c][m] (row length M) instead of part[split][m][c]; needs M % 4 == 0.
// FL ("flat time axis", with RS): the S segments of T samples are contracted as ONE axis of S * T samples cut into
// 32-sample stages, instead of ceil(T / 32) stages per segment -- T = 360 wastes 24 of every 384 samples (6.25 % of the
// MFMAs) otherwise.  Needs T % 4 == 0 (a lane's 4-sample piece then lies inside one segment), T >= 64 and both
// tensors below 2 GB (one descriptor each); a piece's segment and local time are per-lane state, advanced by
// additions; the X samples a tap shift pushes outside their segment are zeroed by the per-value scales, as at the
// segment edges of the per-segment walk.
template <int KS, int MW, int NS, bool RS = false, bool TS = false, bool FL = false>
__global__ __launch_bounds__(256, 1) void gemm_nt_h2w_kernel(GemmNTArgsH a) {
    static_assert(!FL || RS, "the flat walk relies on the row-scaled A path (pieces inside or outside a row)");
    static_assert((MW == 5 && NS == 3) || (MW == 4 && NS == 2) || (MW == 4 && NS == 4),
                  "tile variants: 320 x 192, 256 x 128, 256 x 256");
    static_assert(!TS || (KS == 1 && !RS), "the transposed partial tile is the score contraction's");
    static_assert(KS == 1 || NS == 3, "3 taps use the 3 slots");
    constexpr int NA = 2 * MW;                        // A pieces per thread and stage (32 rows per pass)
    constexpr int NX = 2 * NS;                        // X pieces (slot j, 32-row half u)
    constexpr int NP = NA + NX;
    constexpr int BM = 64 * MW;
    constexpr int BC = KS == 3 ? 64 : 64 * NS;        // X rows of the workgroup tile
    constexpr int AQ = BM + 2;                        // slots of one 8-sample group of the A tile (2 pad rows)
    constexpr int ASLOTS = 4 * AQ;                    // 16-byte slots of one plane of the A tile
    constexpr int PLANE = ASLOTS + NS * 4 * HG_XQ;    // ... of one plane (A + X)
    constexpr int BUF = 2 * PLANE;                    // slots of one stage buffer (2 planes)
    constexpr int TN = MW * NS;                       // MFMAs of one term (one operand-plane pair, one k-step)
    constexpr int NXS = KS == 3 ? 3 : 1;              // tap shifts of the X fetches
#ifdef HG_TRACE
    const long long t_kernel0 = __builtin_readcyclecounter();
    long long t_loop0 = t_kernel0, t_loop1 = t_kernel0;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);      // [2 buffers][2 planes][A slots | X slots]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wc = wave & 1;
    const int nl = lane & 31, h = lane >> 5;

    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % a.tiles_m; id /= a.tiles_m;
    const int tc = id % a.tiles_c; id /= a.tiles_c;
    const int split = id % a.nsplit;
    const int grp = id / a.nsplit;                    // 0 unless grouped
    const int m0 = tm * BM, c0 = tc * BC;
    const int s_first = a.seg ? a.seg[grp] : 0;       // the group's range in `order` (wave-uniform scalars)
    const int s_count = a.seg ? a.seg[grp + 1] - s_first : a.S;
    // segment number of the group's k-th segment
#define HG_SEG(K_) (a.order ? a.order[s_first + (K_)] : s_first + (K_))

    const int cps = (a.T + HG_K - 1) / HG_K;
    const long nchunks = FL ? ((long)a.S * a.T + HG_K - 1) / HG_K : (long)s_count * cps;
    const long q_begin = nchunks * split / a.nsplit;
    const long q_end = nchunks * (split + 1) / a.nsplit;
    const int nst = (int)(q_end - q_begin);
    const int halo = KS == 3 ? a.dil : 0;
    const int a_bytes = (int)(((FL ? (long)(a.S - 1) * a.a_sstride : 0) + (long)(a.M - 1) * a.a_rstride + a.T) * 4);
    const int x_bytes = (int)(((FL ? (long)(a.S - 1) * a.x_sstride : 0) + (long)(a.Cn - 1) * a.x_rstride + a.T) * 4);
    float sa, sa_inv, sx, sx_inv;
    hg_scale_from_amax(bm_amax_load(a.a_amax), sa, sa_inv);
    hg_scale_from_amax(bm_amax_load(a.x_amax), sx, sx_inv);

    // The pieces of this thread: lane -> (row32 = tid >> 3, piece pl = tid & 7 of the 32 samples).  A piece i:
    // row m0 + 32 i + row32.  X piece (j, u) = index NA + 2 j + u: row c0 + 32 u + row32 read at tap shift
    // (j - 1) * dil (KS = 3) or row c0 + 64 j + 32 u + row32 (KS = 1).  Byte offsets inside a segment, all >= 0; a
    // row past M / Cn gets 0x7f000000: past the range of every descriptor (segments span < 0x7f000000 bytes,
    // checked by the host) and small enough that adding a chunk offset cannot wrap (reads 0).
    const int row32 = tid >> 3, pl = tid & 7;
    int offa[NA], offx[NX];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + 32 * i + row32;
        offa[i] = m < a.M ? (m * (int)a.a_rstride + 4 * pl) * 4 : 0x7f000000;
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        const int c = c0 + (KS == 3 ? 0 : 64 * (q >> 1)) + 32 * (q & 1) + row32;
        offx[q] = c < a.Cn ? (c * (int)a.x_rstride + 4 * pl) * 4 : 0x7f000000;
    }
    // LDS byte address inside a plane: 16-byte slot of the 8-sample group (pl >> 1), 8-byte half (pl & 1)
    const int ldsa = ((pl >> 1) * AQ + row32) * 16 + (pl & 1) * 8;                          // piece i: + i * 32 * 16
    const int ldsx = (ASLOTS + (pl >> 1) * HG_XQ + row32) * 16 + (pl & 1) * 8;   // piece (j, u): + (j * 4 * HG_XQ + 32 u) * 16

    f32x16 acc[MW][NS];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 rp[NP];                                      // the staging register set
    // Per-value operand scales: sa / sx, or 0 for a sample outside [0, T) (the dwordx4 fetch of a row's first or
    // last chunk also brings samples of the neighbouring rows; v_mul_legacy: 0 * anything = 0).  Rewritten only
    // when the chunk in the staging set is the first or last one of its segment, or follows one.
    float sva[4], svx[NXS][4];
    float sarow[RS ? NA : 1];                          // RS: scale of the row of A piece i
    if constexpr (RS) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + 32 * i + row32;
            float inv_unused;
            hg_scale_from_amax(m < a.M ? a.a_row_amax[m] : 0.f, sarow[i], inv_unused);
        }
    }
    // (T0_ = first sample of the chunk in the staging set, local to its segment; FL: this lane's piece may already
    // belong to the next segment -- its own local time lt_st is used instead)
#define HG_SET_SCALES(T0_)                                                                        \
    {                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                           \
            const int ta = (FL ? lt_st : (T0_) + 4 * pl) + r;                                     \
            if constexpr (!RS) sva[r] = ta < a.T ? sa : 0.f;                                      \
            _Pragma("unroll") for (int j = 0; j < NXS; ++j) {                                     \
                const int tx = ta + (KS == 3 ? (j - 1) * a.dil : 0);                              \
                svx[j][r] = (tx >= 0 && tx < a.T) ? sx : 0.f;                                     \
            }                                                                                     \
        }                                                                                         \
    }

    int ld_q = 0;
    int ld_s = FL ? 0 : (int)(q_begin / cps);
    int ld_c = FL ? 0 : (int)(q_begin - (long)ld_s * cps);
    // segment descriptors, rebuilt only when the cursor enters a new segment (FL: the whole tensors, once)
    const int seg0 = (FL || nst <= 0) ? 0 : HG_SEG(ld_s);
    __amdgpu_buffer_rsrc_t qa = hg_rsrc(a.a + (long)seg0 * a.a_sstride, a_bytes);
    __amdgpu_buffer_rsrc_t qx = hg_rsrc(a.x + (long)seg0 * a.x_sstride, x_bytes);
    int vn, vx[NXS];                                   // byte offset of the chunk to fetch (X: per tap)
    unsigned vna = 0;                                  // RS: the same for A, out of range when this lane's piece lies past T
    // FL: this lane's piece of the chunk at the cursor -- segment ls, local time lt (a multiple of 4), byte offsets
    // la / lx of (segment start + lt - 4 pl) in A / X; cur_t0 = local time of the chunk's first sample (wave-uniform)
    int ls = 0, lt = 0, la = 0, lx = 0, cur_t0 = 0, lt_st = 0;
    if constexpr (FL) {
        const long f0 = q_begin * HG_K;
        const int s0 = (int)(f0 / a.T);
        cur_t0 = (int)(f0 - (long)s0 * a.T);
        ls = s0;
        lt = cur_t0 + 4 * pl;
        if (lt >= a.T) { lt -= a.T; ++ls; }
        la = (int)(((long)ls * a.a_sstride + lt - 4 * pl) * 4);
        lx = (int)(((long)ls * a.x_sstride + lt - 4 * pl) * 4);
    }
#define HG_CHUNK_OFFSETS()                                                                        \
    {                                                                                             \
        if constexpr (FL) {                                                                       \
            vn = lx;                                                                              \
            vna = ls < a.S ? (unsigned)la : 0x7f000000u;                                          \
        } else {                                                                                  \
            vn = ld_c * (HG_K * 4);                                                               \
            if constexpr (RS) vna = (ld_c * HG_K + 4 * pl < a.T) ? (unsigned)vn : 0x7f000000u;    \
        }                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NXS; ++j) vx[j] = vn + (KS == 3 ? (j - 1) * a.dil * 4 : 0); \
    }
#define HG_ADVANCE()                                                                              \
    if (++ld_q < nst) {                                /* else: stays on the last chunk */        \
        if constexpr (FL) {                                                                       \
            cur_t0 += HG_K;                                                                       \
            if (cur_t0 >= a.T) cur_t0 -= a.T;                                                     \
            lt += HG_K; la += HG_K * 4; lx += HG_K * 4;                                           \
            if (lt >= a.T) {                                                                      \
                lt -= a.T; ++ls;                                                                  \
                la += ((int)a.a_sstride - a.T) * 4;                                               \
                lx += ((int)a.x_sstride - a.T) * 4;                                               \
            }                                                                                     \
        } else if (++ld_c == cps) {                                                               \
            ld_c = 0; ++ld_s;                                                                     \
            const int sg_ = HG_SEG(ld_s);                                                         \
            qa = hg_rsrc(a.a + (long)sg_ * a.a_sstride, a_bytes);                                 \
            qx = hg_rsrc(a.x + (long)sg_ * a.x_sstride, x_bytes);                                 \
        }                                                                                         \
    }
    // fetch of piece I_ of the chunk at the cursor: the whole offset travels in the per-lane address, which the
    // descriptor range-checks (a scalar offset is not), so nothing outside the segment is ever read.  A dwordx4
    // whose first byte lies in front of the segment (first X row, first chunk, tap -dil) comes back as zeros
    // altogether: those lanes re-fetch it sample by sample.
#define HG_Q(I_) ((I_) < NA ? 0 : (I_) - NA)     /* X piece index of staging piece I_ */
#define HG_FETCH(I_)                                                                              \
    {                                                                                             \
        if ((I_) < NA) rp[I_] = hg_ld128(qa, RS ? (int)((unsigned)offa[(I_) < NA ? (I_) : 0] + vna)             \
                                                : offa[(I_) < NA ? (I_) : 0] + vn, 0);            \
        else {                                                                                    \
            const int vo_ = offx[(I_) < NA ? 0 : (I_) - NA] + vx[KS == 3 ? (HG_Q(I_) >> 1) % NXS : 0]; \
            rp[I_] = hg_ld128(qx, vo_, 0);                                                        \
            if (KS == 3 && (I_) == NA && vn == 0) {                                               \
                if (vo_ < 0 && vo_ > -16) {                                                       \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                 \
                        rp[I_][r] = hg_ld32(qx, vo_ + 4 * r < 0 ? 0x7ffffff0 : vo_ + 4 * r);      \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
    }
    // piece I_ of the staging set -> stage buffer at byte address WB_, in three sub-steps that go behind three
    // consecutive MFMAs: split of samples 0-1, of samples 2-3, two 8-byte LDS writes (+ the fetch that refills it)
#define HG_SV(I_) ((I_) < NA ? sva : svx[KS == 3 ? (HG_Q(I_) >> 1) % NXS : 0])
#define HG_S(I_, R_) ((RS && (I_) < NA) ? sarow[(RS && (I_) < NA) ? (I_) : 0] : HG_SV(I_)[R_])
#define HG_SPLIT0(I_) hg_split_pair(__uint_as_float(rp[I_][0]), __uint_as_float(rp[I_][1]), HG_S(I_, 0), HG_S(I_, 1), ph[0], pw[0]);
#define HG_SPLIT1(I_) hg_split_pair(__uint_as_float(rp[I_][2]), __uint_as_float(rp[I_][3]), HG_S(I_, 2), HG_S(I_, 3), ph[1], pw[1]);
#define HG_WRITE(I_, WB_)                                                                         \
    {                                                                                             \
        char* dst_ = (WB_) + ((I_) < NA ? ldsa + (I_) * 32 * 16                                   \
                                        : ldsx + ((((I_) - NA) >> 1) * 4 * HG_XQ + 32 * (((I_) - NA) & 1)) * 16); \
        *reinterpret_cast<u32x2*>(dst_) = u32x2{ph[0], ph[1]};                                    \
        *reinterpret_cast<u32x2*>(dst_ + PLANE * 16) = u32x2{pw[0], pw[1]};                       \
    }
#define HG_STORE(I_, WB_) { HG_SPLIT0(I_) HG_SPLIT1(I_) HG_WRITE(I_, WB_) }
    // the same split in four sub-steps of 2 VALU (hi / lo of samples 0-1, hi / lo of samples 2-3)
#define HG_SPLIT_SUB(I_, Q_)                                                                      \
    {                                                                                             \
        if constexpr ((Q_) == 0) hg_split_hi(__uint_as_float(rp[I_][0]), __uint_as_float(rp[I_][1]), HG_S(I_, 0), HG_S(I_, 1), ph[0]); \
        if constexpr ((Q_) == 1) hg_split_lo(__uint_as_float(rp[I_][0]), __uint_as_float(rp[I_][1]), HG_S(I_, 0), HG_S(I_, 1), ph[0], pw[0]); \
        if constexpr ((Q_) == 2) hg_split_hi(__uint_as_float(rp[I_][2]), __uint_as_float(rp[I_][3]), HG_S(I_, 2), HG_S(I_, 3), ph[1]); \
        if constexpr ((Q_) == 3) hg_split_lo(__uint_as_float(rp[I_][2]), __uint_as_float(rp[I_][3]), HG_S(I_, 2), HG_S(I_, 3), ph[1], pw[1]); \
    }
    unsigned ph[2], pw[2];
    // one operand fragment set, rolling over the k-steps: [plane 0 = hi, 1 = lo]; k-step KK_ of the stage in
    // buffer RB_ (u32x4*): group 2 KK_ + h of the lane's row
#define HG_READ_A(P_, RB_, KK_)                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        af[P_][mt] = __builtin_bit_cast(f16x8, (RB_)[(P_) * PLANE + (2 * (KK_) + h) * AQ + wm * (MW * 32) + mt * 32 + nl]);
#define HG_READ_B(P_, RB_, KK_)                                                                   \
    _Pragma("unroll") for (int j = 0; j < NS; ++j)                                                \
        bf[P_][j] = __builtin_bit_cast(f16x8, (RB_)[(P_) * PLANE + ASLOTS + (j * 4 + 2 * (KK_) + h) * HG_XQ + wc * 32 + nl]);
#define HG_READ_A1(P_, RB_, KK_, MT_)                                                             \
    af[P_][MT_] = __builtin_bit_cast(f16x8, (RB_)[(P_) * PLANE + (2 * (KK_) + h) * AQ + wm * (MW * 32) + (MT_) * 32 + nl]);
#define HG_READ_B1(P_, RB_, KK_, J_)                                                              \
    bf[P_][J_] = __builtin_bit_cast(f16x8, (RB_)[(P_) * PLANE + ASLOTS + ((J_) * 4 + 2 * (KK_) + h) * HG_XQ + wc * 32 + nl]);
    f16x8 af[2][MW], bf[2][NS];

    if (nst > 0) {
        // chunk 0 -> buffer 0, chunk 1 -> staging set
        // HG_MARK: remember where the chunk about to be fetched starts (wave-uniform t0c; FL: this lane's lt_st)
#define HG_MARK() { t0c = FL ? cur_t0 : ld_c * HG_K; if constexpr (FL) lt_st = lt; }
        HG_CHUNK_OFFSETS()
        int t0c;
        HG_MARK()
#pragma unroll
        for (int i = 0; i < NP; ++i) HG_FETCH(i)
        HG_ADVANCE()
        HG_SET_SCALES(t0c)
#pragma unroll
        for (int i = 0; i < NP; ++i) HG_STORE(i, reinterpret_cast<char*>(lds))
        HG_CHUNK_OFFSETS()
        HG_MARK()
#pragma unroll
        for (int i = 0; i < NP; ++i) HG_FETCH(i)
        HG_ADVANCE()
        __syncthreads();
        HG_READ_A(1, lds, 0)
        HG_READ_B(0, lds, 0)
        bool was_edge = true;
#ifdef HG_TRACE
        unsigned long long tr[8], tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        __builtin_amdgcn_sched_barrier(0);
        t_loop0 = __builtin_readcyclecounter();
#endif
        // stage k: MFMAs on buffer k & 1 -- six terms of MW NS MFMAs: k-step 0 lo*hi, hi*hi, hi*lo, k-step 1
        // likewise; the staging set holds chunk k + 1 (fetched during stage k - 1; a repeat of the last chunk in
        // the last stage: split and stored like any other, never read).  Spread over the first five terms, piece
        // by piece: split the piece into buffer (k + 1) & 1, fetch the same piece of chunk k + 2 into its
        // registers.  Barrier before the last term, which covers the first fragment reads of stage k + 1.
        // per-stage bookkeeping (edge scales of the chunk in the staging set, offsets of the chunk to fetch next): for
        // stage 0 here, for stage k + 1 behind the MFMAs of stage k's last term (in front of the loop's MFMA stream it
        // cost ~200 cycles per stage during which the matrix pipe idled: cycle trace, "wait loads")
#define HG_BOOKKEEPING()                                                                          \
        {                                                                                         \
            const bool edge = t0c - halo < 0 || t0c + HG_K + halo > a.T;                          \
            if (edge || was_edge) HG_SET_SCALES(t0c)                                              \
            was_edge = edge;                                                                      \
            HG_CHUNK_OFFSETS()                                                                    \
            HG_MARK()                                                                             \
        }
        HG_BOOKKEEPING()
        for (int k = 0; k < nst; ++k) {
            HG_T(0)
            const u32x4* rb = lds + (k & 1) * BUF;
            const u32x4* nb = lds + ((k + 1) & 1) * BUF;
            char* wb = reinterpret_cast<char*>(lds + ((k + 1) & 1) * BUF);
            HG_T(1)
            // One slot per MFMA, in source order (scheduling fence after each): the MFMA, then at most one
            // fragment read and one sub-step of a piece -- a lone wavefront per SIMD issues one instruction every
            // 8 cycles, so an MFMA (32 cycles) hides 3 more, and a burst of anything else leaves the matrix pipe idle
            // (measured with the cycle trace of scripts/trace_wgrad.py: 16 fetches issued back to back block the
            // wavefront for 2 000 cycles).  Fragments roll through one register set: each is re-read for the next
            // k-step right behind the last MFMA that uses it.
            hg_static_for<6 * TN>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                constexpr int term = n / TN, w = n % TN, mt = w / NS, j = w % NS;
                constexpr int pa = term % 3 == 0 ? 1 : 0, pb = term % 3 == 2 ? 1 : 0;
                if constexpr (n == 5 * TN) {
                    HG_T(4)
                    hg_barrier<MW, NS>(af[0], bf[0]);
                    HG_T(5)
                }
                acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[pa][mt], bf[pb][j], acc[mt][j], 0, 0, 0);
                // fragment reads
                if constexpr (term == 0) {
                    if constexpr (w < MW) HG_READ_A1(0, rb, 0, w)                        // late fragments of k-step 0:
                    else if constexpr (w < MW + NS) HG_READ_B1(1, rb, 0, w - MW)         // A.hi, B.lo
                }
                if constexpr (term == 0 && j == NS - 1 && w >= MW + NS) HG_READ_A1(1, rb, 1, mt)   // A.lo: last use
                if constexpr (term == 1 && w == TN - 1) {
                    // A.lo fragments whose slot was taken by the late reads above
                    hg_static_for<MW>([&](auto mc) __attribute__((always_inline)) {
                        constexpr int m2 = decltype(mc)::value;
                        if constexpr (m2 * NS + NS - 1 < MW + NS) HG_READ_A1(1, rb, 1, m2)
                    });
                }
                if constexpr (term == 1 && mt == MW - 1) HG_READ_B1(0, rb, 1, j)         // B.hi: last use
                if constexpr (term == 2 && j == NS - 1) HG_READ_A1(0, rb, 1, mt)         // A.hi
                if constexpr (term == 2 && mt == MW - 1) HG_READ_B1(1, rb, 1, j)         // B.lo
                if constexpr (term == 5) {                                                // early fragments of k + 1
                    if constexpr (w < MW) HG_READ_A1(1, nb, 0, w)
                    else if constexpr (w < MW + NS) HG_READ_B1(0, nb, 0, w - MW)
                }
                // piece p in FIVE sub-steps (split hi / lo of samples 0-1, hi / lo of samples 2-3: 2 VALU each; the two
                // LDS writes + the refill fetch), the 5 NP sub-steps spread evenly over the 5 TN slots of the first five
                // terms -- whole 4-VALU pairs behind one MFMA overran its 32 cycles (cycle trace: 3 246 cycles for the
                // 75 MFMAs = 2 400 of these terms)
                if constexpr (n < 5 * TN) {
                    constexpr int NSUB = 5 * NP, NSL = 5 * TN;
                    constexpr int q0 = (n * NSUB + NSL - 1) / NSL, q1 = ((n + 1) * NSUB + NSL - 1) / NSL;
                    hg_static_for<q1 - q0>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = q0 + decltype(qc)::value, p = q / 5, sub = q % 5;
                        if constexpr (sub < 4) HG_SPLIT_SUB(p, sub)
                        else {
                            HG_WRITE(p, wb)
                            HG_FETCH(p)
                        }
                    });
                }
                // the next stage's bookkeeping: behind the early fragment reads of the last term (every split and
                // fetch of this stage is issued by then)
                if constexpr (n == 5 * TN + MW + NS) HG_ADVANCE()
                if constexpr (n == 5 * TN + MW + NS + 1) HG_BOOKKEEPING()
                __builtin_amdgcn_sched_barrier(0);
            });
#ifdef HG_TRACE
            HG_T(6)
            tacc[0] += tr[1] - tr[0]; tacc[1] += tr[4] - tr[1]; tacc[4] += tr[5] - tr[4]; tacc[5] += tr[6] - tr[5];
            tacc[7] += 1;
#endif
        }
#ifdef HG_TRACE
        __builtin_amdgcn_sched_barrier(0);
        t_loop1 = __builtin_readcyclecounter();
        if (blockIdx.x < 64 && lane == 0)
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) hg_trace_buf[(blockIdx.x * 4 + wave) * 8 + i_] = (long long)tacc[i_];
#endif
    }
#undef HG_SEG
#undef HG_SET_SCALES
#undef HG_MARK
#undef HG_CHUNK_OFFSETS
#undef HG_ADVANCE
#undef HG_FETCH
#undef HG_Q
#undef HG_STORE
#undef HG_SV
#undef HG_S
#undef HG_SPLIT0
#undef HG_SPLIT_SUB
#undef HG_BOOKKEEPING
#undef HG_SPLIT1
#undef HG_WRITE
#undef HG_READ_A1
#undef HG_READ_B1
#undef HG_READ_A
#undef HG_READ_B

    // partial tile out (inverse scales are exact powers of two): KS = 3: part[split][m][c * 3 + j];
    // KS = 1: part[split][m][c0 + 64 j + ...]
    const long N = (long)a.Cn * KS;
    float* dst = a.part + ((long)grp * a.nsplit + split) * a.M * N;
    if constexpr (RS) {
        // the rows' inverse scales, once per workgroup into the (now free) LDS: 80 rows per lane below
        __syncthreads();
        for (int i = tid; i < BM; i += 256) {
            float s_unused, inv_m = 0.f;
            if (m0 + i < a.M) hg_scale_from_amax(a.a_row_amax[m0 + i], s_unused, inv_m);
            smem[i] = inv_m;
        }
        __syncthreads();
    }
    if constexpr (TS) {
        // element r of a lane: row (r & 3) + 8 (r >> 2) + 4 h of the block -- rows 4 q' .. 4 q' + 3 are r = 4 q .. 4 q + 3
        const float f = sa_inv * sx_inv;
        float* dstT = a.part + ((long)grp * a.nsplit + split) * a.Cn * a.M;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int c = c0 + 64 * j + wc * 32 + nl;
            if (c < a.Cn) {
#pragma unroll
                for (int mt = 0; mt < MW; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = m0 + wm * (MW * 32) + mt * 32 + 8 * q + 4 * h;
                        if (m < a.M)        // M % 4 == 0: the four rows are inside or outside together
                            *reinterpret_cast<float4*>(dstT + (long)c * a.M + m) =
                                float4{acc[mt][j][4 * q] * f, acc[mt][j][4 * q + 1] * f, acc[mt][j][4 * q + 2] * f,
                                       acc[mt][j][4 * q + 3] * f};
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (MW * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < a.M) {
                float f = sa_inv * sx_inv;
                if constexpr (RS) f = smem[m - m0] * sx_inv;
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
#ifdef HG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the partial tile has left the CU
    if (blockIdx.x < 64 && lane == 0) {
        long long* o = hg_trace_buf + (blockIdx.x * 4 + wave) * 8;
        o[2] = t_loop0 - t_kernel0;                               // prologue
        o[3] = t_loop1 - t_loop0;                                 // main loop
        o[6] = (long long)__builtin_readcyclecounter() - t_loop1; // epilogue
    }
#endif
}

// tile family of a shape: 0 = not covered, 1 = 320 x 192 (<KS,5,3>), 2 = 256 x 128 (<1,4,2>)
static int hg_family(int S, int G, int M, int Cn, int T, int KS, int dil, bool ordered) {
    if (KS != 3 && KS != 1) return 0;
    // grouped (order / seg) contractions: the 1x1 tile families, per-segment walk, per-tensor scales
    if ((G != 1 || ordered) && KS != 1) return 0;
    if (dil < 1 || dil > 32 || T < 2 * HG_K) return 0;
    if ((long)S * ((T + HG_K - 1) / HG_K) < 32L * G) return 0;
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

// one workgroup per CU per round (256 CUs), >= 8 stages (of 32 samples) per workgroup
extern "C" int bm_gemm_nt_h2_suggest_splits_grouped(int M, int Cn, int KS, int S, int T, int G);
extern "C" int bm_gemm_nt_h2_suggest_splits(int M, int Cn, int KS, int S, int T) {
    return bm_gemm_nt_h2_suggest_splits_grouped(M, Cn, KS, S, T, 1);
}
// G groups of ~S / G segments each: splits per group so that ~256 workgroups exist, >= 8 stages each
extern "C" int bm_gemm_nt_h2_suggest_splits_grouped(int M, int Cn, int KS, int S, int T, int G) {
    if (G < 1) G = 1;
    const int fam = hg_family(S, G, M, Cn, T, KS, 1, G > 1);
    const int tiles = (fam == 2 ? cdiv(M, 256) * cdiv(Cn, 128) : cdiv(M, 320) * cdiv(Cn, KS == 3 ? 64 : 192)) * G;
    const long chunks = (long)(S / G > 0 ? S / G : 1) * ((T + HG_K - 1) / HG_K);
    long want = 256 / tiles;
    if (want > chunks / 8) want = chunks / 8;
    if (want < 1) want = 1;
    return (int)want;
}

template <int KS, int MW, int NS, bool RS = false, bool TS = false, bool FL = false>
static int launch_gemm_nt_h2w(GemmNTArgsH g, hipStream_t stream, int ngroups = 1) {
    constexpr int PLANE = 4 * (64 * MW + 2) + NS * 4 * HG_XQ;
    const size_t lds = (size_t)2 * 2 * PLANE * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_h2w_kernel<KS, MW, NS, RS, TS, FL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "gemm_nt_h2w: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    g.tiles_m = cdiv(g.M, 64 * MW);
    g.tiles_c = cdiv(g.Cn, KS == 3 ? 64 : 64 * NS);
    const long nblocks = (long)g.tiles_m * g.tiles_c * g.nsplit * ngroups;
    hipLaunchKernelGGL((gemm_nt_h2w_kernel<KS, MW, NS, RS, TS, FL>), dim3((unsigned)nblocks), dim3(256), lds, stream, g);
    return bm_check_launch("gemm_nt_h2w");
}

extern "C" int bm_gemm_nt_x3(const float* a, long a_sstride, long a_rstride, const float* x, long x_sstride,
                             long x_rstride, const int* order, const int* seg, float* part, int S, int G, int M, int Cn,
                             int T, int KS, int dil, int nsplit, void* stream);

static int gemm_nt_h2_impl(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* a_row_amax,
                           const float* x, long x_sstride, long x_rstride, const float* x_amax, float* part, int S,
                           int M, int Cn, int T, int KS, int dil, int nsplit, void* stream,
                           const int* order = nullptr, const int* seg = nullptr, int G = 1) {
    BM_REQUIRE(a && x && part && a_amax && x_amax, "gemm_nt_h2: null pointer");
    BM_REQUIRE(M > 0 && Cn > 0 && T > 0 && nsplit > 0 && S >= 0, "gemm_nt_h2: bad dims");
    BM_REQUIRE(G >= 1 && (G == 1 || seg), "gemm_nt_h2: a grouped call needs seg[]");
    const bool grouped = seg != nullptr || order != nullptr;
    const int fam = hg_family(S, G, M, Cn, T, KS, dil, grouped);
    BM_REQUIRE(fam != 0, "gemm_nt_h2: shape not covered (M=%d Cn=%d T=%d KS=%d dil=%d)", M, Cn, T, KS, dil);
    // 32-bit byte offsets inside a segment: an operand whose segment spans 2 GB or more (2 048 wav2vec2-sized
    // candidates on 8 GPUs) takes the 3 x bf16 entry point, which hands it to the 64-bit-addressed fp32 kernel
    if (((long)(M - 1) * a_rstride + T) * 4 >= 0x7f000000L || ((long)(Cn - 1) * x_rstride + T) * 4 >= 0x7f000000L)
        return bm_gemm_nt_x3(a, a_sstride, a_rstride, x, x_sstride, x_rstride, order, seg, part, S, G, M, Cn, T,
                             KS, dil, nsplit, stream);
    GemmNTArgsH g;
    g.a = a; g.a_sstride = a_sstride; g.a_rstride = a_rstride; g.a_amax = a_amax;
    g.x = x; g.x_sstride = x_sstride; g.x_rstride = x_rstride; g.x_amax = x_amax;
    g.a_row_amax = a_row_amax;
    g.part = part; g.S = S; g.M = M; g.Cn = Cn; g.T = T; g.dil = dil; g.nsplit = nsplit;
    g.order = order; g.seg = seg;
    hipStream_t s = (hipStream_t)stream;
    if (grouped) {          // per-segment walk, per-tensor scales: the two 1x1 tile families
        if (fam == 2) return launch_gemm_nt_h2w<1, 4, 2>(g, s, G);
        return launch_gemm_nt_h2w<1, 5, 3>(g, s, G);
    }
    // per-row scales of A: the 320-row tiles, rows contiguous in time (pieces of 4 samples never straddle a row)
    const bool rs = a_row_amax && fam == 1 && T % 4 == 0 && a_rstride % 4 == 0 && a_sstride % 4 == 0;
    // flat time axis (no per-segment padding of T to a multiple of 32): row-scaled 3-tap weight gradients whose
    // tensors fit one 2 GB descriptor each
    const bool flat = rs && KS == 3 && T % HG_K != 0 && T >= 2 * HG_K && S > 1 &&
                      ((long)(S - 1) * a_sstride + (long)(M - 1) * a_rstride + T) * 4 < 0x7f000000L &&
                      ((long)(S - 1) * x_sstride + (long)(Cn - 1) * x_rstride + T) * 4 < 0x7f000000L &&
                      a_sstride >= T && x_sstride >= T && x_sstride % 4 == 0;
    if (flat) return launch_gemm_nt_h2w<3, 5, 3, true, false, true>(g, s);
    if (KS == 3) return rs ? launch_gemm_nt_h2w<3, 5, 3, true>(g, s) : launch_gemm_nt_h2w<3, 5, 3>(g, s);
    if (fam == 2) return launch_gemm_nt_h2w<1, 4, 2>(g, s);
    return rs ? launch_gemm_nt_h2w<1, 5, 3, true>(g, s) : launch_gemm_nt_h2w<1, 5, 3>(g, s);
}

// part[split][m][c*KS + j] for one group of S consecutive segments (same contract as bm_gemm_nt without
// order / seg); a_amax / x_amax: device pointers to max|a|, max|x| (bm_amax).  Only shapes
// bm_gemm_nt_h2_covers() accepts.
extern "C" int bm_gemm_nt_h2(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* x,
                             long x_sstride, long x_rstride, const float* x_amax, float* part, int S, int M, int Cn,
                             int T, int KS, int dil, int nsplit, void* stream) {
    return gemm_nt_h2_impl(a, a_sstride, a_rstride, a_amax, nullptr, x, x_sstride, x_rstride, x_amax, part, S, M, Cn, T,
                           KS, dil, nsplit, stream);
}

// The same with `a_row_amax` ([M] floats, device; nullable): per-row max |a| -- every row of A (one gradient channel =
// one row of dW) then carries its own scale, so a channel far below its tensor's maximum keeps full accuracy.
// Used by the 320-row tile family when T % 4 == 0; ignored otherwise (the per-tensor scale applies).
extern "C" int bm_gemm_nt_h2_rows(const float* a, long a_sstride, long a_rstride, const float* a_amax,
                                  const float* a_row_amax, const float* x, long x_sstride, long x_rstride,
                                  const float* x_amax, float* part, int S, int M, int Cn, int T, int KS, int dil,
                                  int nsplit, void* stream) {
    return gemm_nt_h2_impl(a, a_sstride, a_rstride, a_amax, a_row_amax, x, x_sstride, x_rstride, x_amax, part, S, M, Cn,
                           T, KS, dil, nsplit, stream);
}

// Grouped form: part[g][split][m][c] = sum over the segments order[seg[g] .. seg[g + 1]) (order nullable = identity)
// and the split's share of their samples.  KS = 1 only (the per-(layout, subject) / per-subject weight gradients).
extern "C" int bm_gemm_nt_h2_grouped(const float* a, long a_sstride, long a_rstride, const float* a_amax, const float* x,
                                     long x_sstride, long x_rstride, const float* x_amax, const int* order,
                                     const int* seg, float* part, int S, int G, int M, int Cn, int T, int nsplit,
                                     void* stream) {
    return gemm_nt_h2_impl(a, a_sstride, a_rstride, a_amax, nullptr, x, x_sstride, x_rstride, x_amax, part, S, M, Cn, T,
                           1, 1, nsplit, stream, order, seg, G);
}

// ---- ClipLoss score contraction (bm/losses.py:94, the einsum "bct,oct->bo" before the candidate norms) ----
// part[split][b][o] = sum over split `split` of the K axis of est[b][k] * cand[o][k], est [B][K], cand [Bc][K] dense.
// 256 x 256 tiles with the CANDIDATES as the row operand and the tile stored transposed (kernel <1,4,4,TS>).
// column tile of the estimates: 256 (kernel <1,4,4>) when that still leaves a workgroup >= 16 stages of K, else 128
// (<1,4,2>: twice the workgroups per split, half the partial-tile bytes -- the F = 120 shapes, whose K is short)
static int clip_scores_wide(int B, int Bc, long K) {
    const long tiles = (long)cdiv(Bc, 256) * cdiv(B, 256);
    const long chunks = (K + HG_K - 1) / HG_K;
    long splits = 256 / tiles;
    if (splits < 1) splits = 1;
    return chunks / splits >= 16;
}

extern "C" int bm_clip_scores_h2_covers(int B, int Bc, long K) {
    if (B <= 0 || Bc <= 0 || Bc % 4 != 0 || K < 16 * HG_K || K >= 0x7fffffffL) return 0;
    if (((long)(Bc - 1) * K + K) * 4 >= 0x7f000000L || ((long)(B - 1) * K + K) * 4 >= 0x7f000000L) return 0;
    // padded rows / columns are wasted MFMA work
    const int bc = clip_scores_wide(B, Bc, K) ? 256 : 128;
    return (long)cdiv(Bc, 256) * 256 * cdiv(B, bc) * bc * 2 <= (long)B * Bc * 3;
}

extern "C" int bm_clip_scores_h2_suggest_splits(int B, int Bc, long K) {
    const int bc = clip_scores_wide(B, Bc, K) ? 256 : 128;
    const int tiles = cdiv(Bc, 256) * cdiv(B, bc);
    const long chunks = (K + HG_K - 1) / HG_K;
    long want = 256 / tiles;
    if (want > chunks / 4) want = chunks / 4;
    return (int)(want < 1 ? 1 : want);
}

extern "C" int bm_clip_scores_h2(const float* est, const float* est_amax, const float* cand, const float* cand_amax,
                                 float* part, int B, int Bc, long K, int nsplit, void* stream) {
    BM_REQUIRE(est && cand && part && est_amax && cand_amax, "clip_scores_h2: null pointer");
    BM_REQUIRE(nsplit > 0, "clip_scores_h2: bad split count");
    BM_REQUIRE(bm_clip_scores_h2_covers(B, Bc, K), "clip_scores_h2: shape not covered (B=%d Bc=%d K=%ld)", B, Bc, K);
    GemmNTArgsH g;
    g.a = cand; g.a_sstride = 0; g.a_rstride = K; g.a_amax = cand_amax; g.a_row_amax = nullptr;
    g.x = est; g.x_sstride = 0; g.x_rstride = K; g.x_amax = est_amax;
    g.part = part; g.S = 1; g.M = Bc; g.Cn = B; g.T = (int)K; g.dil = 1; g.nsplit = nsplit;
    g.order = nullptr; g.seg = nullptr;
    if (clip_scores_wide(B, Bc, K)) return launch_gemm_nt_h2w<1, 4, 4, false, true>(g, (hipStream_t)stream);
    return launch_gemm_nt_h2w<1, 4, 2, false, true>(g, (hipStream_t)stream);
}
