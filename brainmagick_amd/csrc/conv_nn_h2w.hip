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
// Tile economy: ONE workgroup of four wavefronts per CU, one wavefront per SIMD, wavefront tile (32 MW) x 96 as
// MW x 3 MFMA accumulators, workgroup tile (64 MW) x 192, MW in {5, 4, 2}; two operand planes: stage =
// (16-channel chunk, tap) = 9 MW MFMAs per wavefront.  LDS: X [2 buffers][2 planes][2 groups][256 columns] x 16 B
// (the weights do not pass through it, see below).
// Packed weights: [g][chunk32][tap][plane][4 groups][Mpad] 16-byte slots (8 f16 channels), followed by the
// per-row inverse scales [G][Mpad] fp32 (bm_pack_weights_h2).
#include "conv_h2_common.h"

// ---------------------------------------------------------------------------------------------------------
// Production main loop (round 6): the WEIGHT operand goes from L2 straight into MFMA fragment registers.
//
// The packed layout [chunk32][tap][plane][4 groups][Mpad] x 16 B already is fragment order -- lane (nl, h) of row
// block mt wants the slot (group = 2 (c16 & 1) + h, row = m0 + wm 32 MW + 32 mt + nl) -- so a wavefront fetches a
// fragment with ONE buffer_load_dwordx4 (two 512-byte runs), one stage ahead, into a second register set: no LDS-DMA
// copy, no LDS write, no ds_read for A (the stage of conv_nn_h2d_kernel below: 5 copies + 10 reads per wavefront, and
// a workgroup barrier per STAGE because the slab is shared).  LDS holds only the two X window buffers, and the
// workgroup meets at one barrier per 16-channel chunk.  B fragments are double-buffered too (read during the
// previous stage), so a stage is 3 MW NW MFMAs on operands that are all in registers when it starts:
//     term 0: A.lo x B.hi,  term 1: A.hi x B.lo,  term 2: A.hi x B.hi.
// What scripts/micro/wino_stage_probe.hip (arm "rega", profiles/r6_rega_probe_*.txt) measured on the way here:
//   * the four wavefronts leave a barrier in lockstep; ten 1 KB loads each in consecutive MFMA slots hand the CU's
//     one texture-address path twice what it takes per slot and the stage costs 1 760 cycles (LDS-DMA: 1 920) --
//     ONE A load every FOURTH slot: 1 640, and the barrier costs nothing any more (it was absorbing that jitter);
//   * the window's split two VALU at a time (a half pair per free slot, one LDS write per slot): 1 616;
//   * a third A register set (two stages ahead), LDS flags instead of the barrier, window loads in two batches: no
//     gain / slower, not built.
// Every load is compiler-visible (raw buffer loads; hipcc counts vmcnt itself and waits at the first use).
// Slots: the n-th MFMA of a stage is followed by at most one A load (n % 4 == 0) or by its share of the stage's
// ordered item list (window loads / split halves / LDS writes / the barrier / the B reads of the next stage),
// spread evenly over the other slots.
// ---------------------------------------------------------------------------------------------------------
#ifdef HG_TRACE
__device__ unsigned ch_trace_buf[64 * 4 * 24];
extern "C" int bm_debug_trace_read_conv(unsigned* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_trace_buf), sizeof(unsigned) * 64 * 4 * 24);
}
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t h2_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// the two halves of ch_split_pair as separate statements (2 VALU each): they sit in different MFMA slots
__device__ __forceinline__ void ch_split_hi(float x0, float x1, float s, unsigned& hi) {
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(hi) : "v"(x0), "v"(x1), "v"(s));
}
__device__ __forceinline__ void ch_split_lo(float x0, float x1, float s, unsigned hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo) : "v"(hi), "v"(x0), "v"(x1), "v"(s));
}

// item kinds of a stage's list
enum { H2R_NONE = 0, H2R_WLOAD, H2R_SPLIT, H2R_WRITE, H2R_BAR, H2R_BREAD };
struct H2RItem { int kind, arg; };
// Item i of the list of stage (KS, tap J); NB = 2 NW B reads.
//   3 taps, J = 0: the 16 loads of the next chunk's window, the B reads of tap 1;
//           J = 1: the B reads of tap 2 (IN FRONT of the barrier: they are the last reads of this chunk's buffer
//                  pair before the next chunk overwrites the other buffer, and only a barrier behind them orders
//                  them against that write), the window's 16 split halves with its 4 LDS writes (a group's two
//                  behind its 8 halves), the barrier;      J = 2: the B reads of the next chunk's tap 0;
//   1x1: split half, split half, then the two registers of the pair refilled with the window after next, x 8, the 4
//        writes in between, the barrier, the B reads of the next chunk.
template <int KS, int J, int NB>
constexpr int h2r_nitems() {
    if (KS == 1) return 32 + 4 + 1 + NB;
    return J == 0 ? 16 + NB : J == 1 ? 16 + 4 + 1 + NB : NB;
}
template <int KS, int J, int NB>
constexpr H2RItem h2r_item(int i) {
    if (KS == 1) {
        // per unit u: S(2u) S(2u+1) L(2u) L(2u+1); writes 0,1 behind unit 3, writes 2,3 behind unit 7
        if (i < 16) return (i & 3) < 2 ? H2RItem{H2R_SPLIT, (i >> 2) * 2 + (i & 1)} : H2RItem{H2R_WLOAD, (i >> 2) * 2 + (i & 1)};
        if (i < 18) return H2RItem{H2R_WRITE, i - 16};
        if (i < 34) { const int k = i - 18; return (k & 3) < 2 ? H2RItem{H2R_SPLIT, 8 + (k >> 2) * 2 + (k & 1)} : H2RItem{H2R_WLOAD, 8 + (k >> 2) * 2 + (k & 1)}; }
        if (i < 36) return H2RItem{H2R_WRITE, i - 34 + 2};
        if (i == 36) return H2RItem{H2R_BAR, 0};
        return H2RItem{H2R_BREAD, i - 37};
    }
    if (J == 0) return i < 16 ? H2RItem{H2R_WLOAD, i} : H2RItem{H2R_BREAD, i - 16};
    if (J == 1) {
        if (i < NB) return H2RItem{H2R_BREAD, i};
        i -= NB;
        if (i < 8) return H2RItem{H2R_SPLIT, i};
        if (i < 10) return H2RItem{H2R_WRITE, i - 8};
        if (i < 18) return H2RItem{H2R_SPLIT, i - 2};
        if (i < 20) return H2RItem{H2R_WRITE, i - 18 + 2};
        return H2RItem{H2R_BAR, 0};
    }
    return H2RItem{H2R_BREAD, i};
}

// RESK: the epilogue of y_out = conv + bias + residual alone (conv_h2_common.h: h2_residual_add), a kernel of its own
template <int KS, int MW, bool RESK = false>
__global__ __launch_bounds__(256, 1) void conv_nn_h2w_kernel(ConvH2Args args) {
    const ConvNNArgs& a = args.c;
#ifdef HG_TRACE
    const unsigned t_kernel0 = (unsigned)__builtin_readcyclecounter();
#endif
    constexpr int NW = 3;
    constexpr int HBM = 64 * MW;                      // rows of the workgroup tile
    constexpr int TN = MW * NW, NS = 3 * TN;          // MFMAs of a term / of a stage
    constexpr int NA = 2 * MW, NB = 2 * NW;           // A loads / B reads of a stage
    constexpr int NF = NS - (NS + 3) / 4;             // free slots (n % 4 != 0)
    static_assert(4 * (NA - 1) < NS, "the A loads of a stage need a slot each");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem);       // [2 buffers][2 planes][2 groups][HXWP]
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
    const int n16 = 2 * a.nchunk;                     // 16-channel chunks, an even number; channels past Cin read 0
    float sx, sx_inv;
    h2_scale_from_amax(bm_amax_load(args.x_amax), sx, sx_inv);

    // this group's packed planes [chunk32][tap][plane][4][Mpad] x 16 B behind one descriptor; a lane's slot inside a
    // (stage, plane): (h Mpad + row) x 16, row block mt = an immediate (512 mt), stage and plane = the scalar offset
    const long gslots = (long)a.nchunk * KS * 8 * a.Mpad;
    const __amdgpu_buffer_rsrc_t wr = h2_rsrc(reinterpret_cast<const u32x4*>(a.wp) + (long)g * gslots, (unsigned)(gslots * 16));
    const int wvoff = (h * a.Mpad + m0 + wm * (MW * 32) + nl) * 16;
    const int pstride = 4 * a.Mpad * 16;              // bytes between the two planes of a stage
    // input window of this segment through a bounds-checked descriptor: channels past Cin read 0; thread `tid`
    // stages window column tid (both 8-channel groups); columns outside [0, T) or past the window get an offset
    // that stays out of range for every channel -> they read as 0 (conv zero padding)
    const __amdgpu_buffer_rsrc_t xr = h2_rsrc(a.x + (long)b * a.x_bstride, (unsigned)(a.Cin * a.T * 4));
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
    unsigned ph[8], pw[8];                            // the window being split: 8 pairs, planes hi / lo
    f16x8 af[2][2][MW], bf[2][2][NW];                 // [register set][plane 0 = hi, 1 = lo][block]

    // A fragment K_ (lo plane first: term 0 uses it) of the stage at scalar byte offset SOFF_ into set SET_
#define H2R_LOAD_A(SET_, K_, SOFF_)                                                                       \
    af[SET_][1 - (K_) / MW][(K_) % MW] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(  \
        wr, wvoff + ((K_) % MW) * 512, (SOFF_) + (1 - (K_) / MW) * pstride, 0));
    // scalar byte offset of stage (C16_, J_): ((chunk32 KS + tap) 8 + 2 (c16 & 1)) Mpad slots
#define H2R_SOFF(C16_, J_) (((((C16_) >> 1) * KS + (J_)) * 8 + ((C16_) & 1) * 2) * a.Mpad * 16)
    // window value R_ (channel R_ of 16) of chunk C16_ for this thread's column
#define H2R_LOAD_X(R_, CB_) xreg[R_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (CB_) + (R_) * crow, 0, 0));
    const u32x4* xrd = Xs + h * HXWP + wn * (NW * 32) + nl;      // B fragment (plane p, block nt): + p 2 HXWP + 32 nt + tap shift
    u32x4* xwr = Xs + tid;                                       // this thread's column: + (plane 2 + group) HXWP

    // prologue: the window of chunk 0 (requested first: it comes from HBM), A of stage 0; split + written to X
    // buffer 0; 1x1: the window of chunk 1 requested behind it; barrier; B of stage 0
    h2_static_for<16>([&](auto rc) __attribute__((always_inline)) { H2R_LOAD_X(decltype(rc)::value, xoff0) });
    h2_static_for<NA>([&](auto kc) __attribute__((always_inline)) { H2R_LOAD_A(0, decltype(kc)::value, 0) });
    h2_static_for<8>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        ch_split_hi(xreg[2 * u], xreg[2 * u + 1], sx, ph[u]);
        ch_split_lo(xreg[2 * u], xreg[2 * u + 1], sx, ph[u], pw[u]);
    });
    xwr[(0 * 2 + 0) * HXWP] = u32x4{ph[0], ph[1], ph[2], ph[3]};
    xwr[(1 * 2 + 0) * HXWP] = u32x4{pw[0], pw[1], pw[2], pw[3]};
    xwr[(0 * 2 + 1) * HXWP] = u32x4{ph[4], ph[5], ph[6], ph[7]};
    xwr[(1 * 2 + 1) * HXWP] = u32x4{pw[4], pw[5], pw[6], pw[7]};
    if (KS == 1) h2_static_for<16>([&](auto rc) __attribute__((always_inline)) { H2R_LOAD_X(decltype(rc)::value, 16 * crow + xoff0) });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    h2_static_for<NB>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        bf[0][i / NW][i % NW] = __builtin_bit_cast(f16x8, xrd[(i / NW) * 2 * HXWP + (i % NW) * 32]);
    });

#ifdef HG_TRACE
    unsigned tr[8], tacc[3][8];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) tacc[i][k] = 0;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned t_loop0 = (unsigned)__builtin_readcyclecounter();
#endif
    // stage (c16, tap J) on register set PAR: 3 MW NW MFMAs, the A loads and the B reads of the NEXT stage into set
    // PAR ^ 1, the window work of its tap
    auto stage = [&](int c16, auto jc, auto pc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, par = decltype(pc)::value;
        CH_T(0)
        // next stage (clamped to the last one: its A is fetched again, harmlessly), its A offset, X buffer and tap
        const bool last_chunk = c16 + 1 >= n16;
        const int nc16 = (j == KS - 1 && !last_chunk) ? c16 + 1 : c16;
        const int soff = (j == KS - 1) ? H2R_SOFF(nc16, last_chunk ? KS - 1 : 0) : H2R_SOFF(c16, j + 1);
        const u32x4* xnext = xrd + ((j == KS - 1 ? c16 + 1 : c16) & 1) * HXSLAB + (j == KS - 1 ? 0 : (j + 1) * a.dil);
        // the window this stage requests (3 taps: chunk c16 + 1, in tap 0; 1x1: chunk c16 + 2) / splits (chunk c16 + 1)
        const int cb = (c16 + (KS == 1 ? 2 : 1)) * 16 * crow + xoff0;
        u32x4* xw = xwr + ((c16 + 1) & 1) * HXSLAB;
        h2_static_for<NS>([&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            constexpr int term = n / TN, w = n % TN, mt = w / NW, nt = w % NW;
            constexpr int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[par][pa][mt], bf[par][pb][nt], acc[mt][nt], 0, 0, 0);
            if constexpr (n % 4 == 0) {
                if constexpr (n / 4 < NA) H2R_LOAD_A(par ^ 1, n / 4, soff)
            } else {
                constexpr int f = (n / 4) * 3 + n % 4 - 1;
                constexpr int L = h2r_nitems<KS, j, NB>();
                constexpr int i0 = (f * L + NF - 1) / NF, i1 = ((f + 1) * L + NF - 1) / NF;
                h2_static_for<i1 - i0>([&](auto ic) __attribute__((always_inline)) {
                    constexpr H2RItem it = h2r_item<KS, j, NB>(i0 + decltype(ic)::value);
                    if constexpr (it.kind == H2R_WLOAD) {
                        H2R_LOAD_X(it.arg, cb)
                    } else if constexpr (it.kind == H2R_SPLIT) {
                        constexpr int u = it.arg / 2;
                        if constexpr (it.arg % 2 == 0) ch_split_hi(xreg[2 * u], xreg[2 * u + 1], sx, ph[u]);
                        else ch_split_lo(xreg[2 * u], xreg[2 * u + 1], sx, ph[u], pw[u]);
                    } else if constexpr (it.kind == H2R_WRITE) {
                        constexpr int kg = it.arg / 2, plane = it.arg % 2;
                        if constexpr (plane == 0) xw[(0 * 2 + kg) * HXWP] = u32x4{ph[4 * kg], ph[4 * kg + 1], ph[4 * kg + 2], ph[4 * kg + 3]};
                        else xw[(1 * 2 + kg) * HXWP] = u32x4{pw[4 * kg], pw[4 * kg + 1], pw[4 * kg + 2], pw[4 * kg + 3]};
                    } else if constexpr (it.kind == H2R_BAR) {
                        // every wavefront's columns of the next window are in LDS (and everybody is done reading the
                        // buffer that the NEXT chunk's window will overwrite)
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    } else if constexpr (it.kind == H2R_BREAD) {
                        constexpr int p = it.arg / NW, q = it.arg % NW;
                        bf[par ^ 1][p][q] = __builtin_bit_cast(f16x8, xnext[p * 2 * HXWP + q * 32]);
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        CH_T(4)
#ifdef HG_TRACE
        tacc[j][0] += tr[4] - tr[0]; tacc[j][7] += 1;
#endif
    };
    // a chunk whose first stage runs on register set CP (3 taps: the sets alternate inside the chunk)
    auto chunk = [&](int c16, auto cpc) __attribute__((always_inline)) {
        constexpr int cp = decltype(cpc)::value;
        h2_static_for<KS>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            stage(c16, jc, std::integral_constant<int, (cp + j) & 1>{});
        });
    };
    // ONE loop over chunk pairs and nothing behind it (a peeled tail chunk made hipcc rotate the accumulators through
    // copies and spill them in the 1x1 kernels): an odd chunk count runs one more chunk on zeros -- the packed planes
    // are padded to whole 32-channel chunks, the window reads past Cin as 0
    for (int c16 = 0; c16 < n16; c16 += 2) {
        chunk(c16, std::integral_constant<int, 0>{});
        chunk(c16 + 1, std::integral_constant<int, KS & 1>{});
    }
#ifdef HG_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const unsigned t_loop1 = (unsigned)__builtin_readcyclecounter();
    if (blockIdx.x < 64 && lane == 0)
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 8; ++k) ch_trace_buf[((blockIdx.x * 4 + wave) * 3 + i) * 8 + k] = tacc[i][k];
#endif
    // the (clamped, unused) loads of the last stage have landed and every wavefront is done with the X buffers
    // before the LDS is re-used by the epilogue
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#undef H2R_LOAD_A
#undef H2R_SOFF
#undef H2R_LOAD_X

#ifdef HG_TRACE
    h2_tile_epilogue<MW, RESK>(args, acc, smem, b, g, m0, n0, ntile, sx_inv, tid, lane, wm, wn, nl, h, ch_trace_buf);
#else
    h2_tile_epilogue<MW, RESK>(args, acc, smem, b, g, m0, n0, ntile, sx_inv, tid, lane, wm, wn, nl, h);
#endif
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
    // BM_CONV_LDSDMA=1: the round-2..5 main loop (weight slabs by LDS-DMA, a barrier per stage; conv_nn_h2d.hip), for A/B runs
    static const bool ldsdma = [] { const char* e = getenv("BM_CONV_LDSDMA"); return e && e[0] == '1'; }();
    if (!ldsdma) lds = (size_t)(2 * HXSLAB) * 16 > lds_ep ? (size_t)(2 * HXSLAB) * 16 : lds_ep;
    // y_out = conv + bias + residual and nothing else (the data-gradient convs of the residual layers, all 3-tap): the
    // kernel with the two-phase residual epilogue
    const ConvNNArgs& c = args.c;
    const bool resk = KS == 3 && !ldsdma && c.res && c.y_out && !c.y_pre && !c.ep_scale && !c.stats && c.act == BM_ACT_NONE;
    static bool attr_set = false;
    if (!ldsdma && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_h2w_kernel<KS, MW, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && KS == 3)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_h2w_kernel<KS, MW, KS == 3>),
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
    if (ldsdma) {
        if (int rc = bm_launch_conv_nn_h2d(args, KS, MW, lds, (unsigned)nblocks, stream)) return rc;
    } else if (resk)
        hipLaunchKernelGGL((conv_nn_h2w_kernel<KS, MW, KS == 3>), dim3((unsigned)nblocks), dim3(256), lds, stream, args);
    else
        hipLaunchKernelGGL((conv_nn_h2w_kernel<KS, MW, false>), dim3((unsigned)nblocks), dim3(256), lds, stream, args);
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
