// What would the main loop of a Winograd F(2,3) conv cost on this datapath, in the BEST case -- every operand a
// ready-made f16 plane staged by LDS-DMA, nothing transformed or split in the kernel -- next to the loop of today's
// direct 3-tap conv (conv_nn_h2w.hip)?  Both variants are built from the production kernel's ingredients, on RANDOM
// operands (the matrix pipe is power-limited: zeros would flatter both):
//   * one workgroup of four wavefronts per CU (one per SIMD), 15 accumulator blocks of 32 x 32 per wavefront (240 AGPRs),
//     45 x v_mfma_f32_32x32x16_f16 per stage, one instruction slot behind every MFMA (scheduling fence after each);
//   * 16 fragment reads (ds_read_b128) per stage from the LDS buffer of the current stage;
//   * the operand bytes of stage s + 1 by global_load_lds (1 KB per instruction), drained and a workgroup barrier
//     at the end of the stage;
//   * an epilogue that writes the tile's fp32 outputs.
//
//   direct   : 60 stages per 320 x 192 tile (20 channel chunks x 3 taps); per stage and wavefront 5 weight copies
//              (20.5 KB per workgroup, an L2-resident 1.2 MB tensor) and, in one stage of three, the input window
//              through registers (16 dword loads + 8 split pairs of 4 VALU + 4 LDS writes per thread); stores 245 KB;
//   winograd : 20 stages per 160 x 192 HALF tile, wavefront = component (the same 5 x 3 block economy); per stage and
//              wavefront 10 copies of transformed weights (41 KB per workgroup, L2-resident 1.6 MB) + 6 copies of
//              transformed input planes (24.6 KB per workgroup, streamed from HBM); the four components are combined
//              through LDS before the stores (60 ds_write_b128 + 60 ds_read_b128 + 240 VALU per lane); stores 123 KB;
//              TWICE as many tiles for the same outputs.
//   rega     : (round 6) the direct conv with the WEIGHT operand off the LDS path: every wavefront fetches its own
//              A fragments (2 planes x 5 row blocks x 1 KB) with buffer_load_dwordx4 straight from L2 into a second
//              fragment register set, one stage ahead (the packed layout [plane][group][Mpad] x 16 B already is
//              fragment order: a lane's slot is h * Mpad + row); LDS holds only the two X window buffers, so the
//              workgroup meets at ONE barrier per 16-channel chunk (3 stages) instead of one per stage; B fragments
//              double-buffered as well (read during the previous stage).  Same MFMA interleave, same window path, same
//              epilogue.  `stagger` > 0: workgroup group (blockIdx / 8) % 4 starts stagger x group cycles late, so that
//              the groups' store drains do not meet on the HBM write path.
// Reported: cycles per stage, cycles per tile, and the time of one conv-sized launch (B = 256: 512 tiles / 1 024 half
// tiles on 256 CUs).  The production conv takes 234 us for the same work.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/wino_stage_probe.hip -o scripts/micro/bin/wino_stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void split_pair(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(x0), "v"(x1), "v"(s));
}

// the two halves of split_pair as separate statements (2 VALU each), so that they can sit in different MFMA slots
__device__ __forceinline__ void split_hi(float x0, float x1, float s, unsigned& hi) {
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(hi) : "v"(x0), "v"(x1), "v"(s));
}
__device__ __forceinline__ void split_lo(float x0, float x1, float s, unsigned hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(hi), "v"(x0), "v"(x1), "v"(s));
}

constexpr int NM = 45;                       // MFMAs per stage and wavefront (5 x 3 blocks x 3 products)
// LDS (16-byte slots).  direct: A 3 buffers x 1 280 (weights of stage s + 2 in flight, like production) + X 2 x 1 024;
// winograd: U 2 buffers x 2 560 (41 KB per stage: a third buffer does not fit) + V 3 buffers x 1 536 (the HBM stream gets
// the two-stage lead) = 155.6 KB of the 160 KB.
constexpr int D_A = 1280, D_X = 1024, D_TOTAL = 3 * D_A + 2 * D_X;
constexpr int W_U = 2560, W_V = 1536, W_TOTAL = 2 * W_U + 3 * W_V;

#define DMA(SRC_, DST_) \
    __builtin_amdgcn_global_load_lds((const void*)(SRC_), (__attribute__((address_space(3))) void*)(DST_), 16, 0, 0);

template <bool WINO>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* __restrict__ wsrc, long wslots, const u32x4* __restrict__ xsrc,
                                                long xslots, float* __restrict__ out, long long* __restrict__ clk,
                                                int tiles_per_wg, int nstage) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[5][3];
    f16x8 af[2][5], bf[2][3];
    float xreg[16];
    unsigned ph[4], pw[4];
    for (int i = 0; i < 16; ++i) xreg[i] = 1.f;
    for (int i = 0; i < 5; ++i) af[0][i] = af[1][i] = f16x8{1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < 3; ++i) bf[0][i] = bf[1][i] = f16x8{1, 1, 2, 2, 3, 3, 4, 4};
    long long t_loop = 0, t_tile = 0;
    const long wper = 64L * 16;                                    // slots one wavefront walks per stage (<= 10 copies)
    long wpos = (wave * 131L * 64) % (wslots - wper);
    const long xshare = xslots / gridDim.x;
    const long xbeg = (long)blockIdx.x * xshare + wave * (xshare / 4);
    const long xend = xbeg + xshare / 4 - 64 * 48;                 // a stage reaches up to 1 024 slots past its position
    long xpos = xbeg;
    const float* xf = reinterpret_cast<const float*>(xsrc);
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
        const long long t0 = clock64();
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
        // prologue: the operands of the first stage(s)
        if (WINO) {
            for (int i = 0; i < 10; ++i) DMA(wsrc + wpos + lane + 64 * i, lds + (wave * 10 + i) * 64)
            for (int b = 0; b < 2; ++b)
                for (int i = 0; i < 6; ++i) DMA(xsrc + xpos + lane + 64 * (6 * b + i), lds + 2 * W_U + b * W_V + (wave * 6 + i) * 64)
        } else {
            for (int b = 0; b < 2; ++b)
                for (int i = 0; i < 5; ++i) DMA(wsrc + wpos + lane + 64 * (5 * b + i), lds + b * D_A + (wave * 5 + i) * 64)
            for (int r = 0; r < 16; ++r) xreg[r] = xf[xpos * 4 + (long)r * 256 + tid];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const long long t1 = clock64();
        int a3 = 0, v3 = 0, s = 0;                                 // s % 3 (A / V buffer rings), stage counter
        // one stage with a COMPILE-TIME tap j (direct; like the production kernel's unrolled tap loop), winograd: j unused
        auto stage = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int a_cur = a3 * D_A, a_nxt2 = (a3 == 0 ? 2 : a3 - 1) * D_A;          // (s + 2) % 3
            const int u_cur = (s & 1) * W_U, u_nxt = W_U - u_cur;
            const int v_cur = 2 * W_U + v3 * W_V, v_nxt2 = 2 * W_U + (v3 == 0 ? 2 : v3 - 1) * W_V;
            const int xb = (s / 3) & 1;
            const int x_cur = 3 * D_A + xb * D_X, x_nxt = 3 * D_A + (1 - xb) * D_X;
            const u32x4* wp = wsrc + wpos + lane;
            const u32x4* xp = xsrc + xpos + lane;
            const float* xfp = xf + xpos * 4 + tid;
            static_for<NM>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                constexpr int blk = n % 15, mt = blk / 3, nt = blk % 3, term = n / 15;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[term & 1][mt], bf[(term + 1) & 1][nt], acc[mt][nt], 0, 0, 0);
                // 16 fragment reads, one per slot
                if constexpr (n < 10)
                    af[(n / 5) & 1][n % 5] = __builtin_bit_cast(f16x8, lds[(WINO ? u_cur + wave * 640 : a_cur) + (n * 64 + lane)]);
                else if constexpr (n < 16)
                    bf[(n / 3) & 1][n % 3] = __builtin_bit_cast(f16x8, lds[(WINO ? v_cur + wave * 384 : x_cur) + ((n - 10) * 64 + lane)]);
                if constexpr (WINO) {
                    // transformed weights of stage s + 1 (10 copies, L2-resident), planes of stage s + 2 (6 copies, HBM)
                    if constexpr (n >= 16 && n < 26) DMA(wp + 64 * (n - 16), lds + u_nxt + (wave * 10 + n - 16) * 64)
                    if constexpr (n >= 26 && n < 32) DMA(xp + 64 * (n - 26), lds + v_nxt2 + (wave * 6 + n - 26) * 64)
                } else {
                    // weights of stage s + 2 (5 copies); the window of the next chunk: 16 loads in the first tap's stage,
                    // split pair by pair into the other X buffer in the last tap's stage
                    if constexpr (n >= 16 && n < 21) DMA(wp + 64 * (n - 16), lds + a_nxt2 + (wave * 5 + n - 16) * 64)
                    if constexpr (j == 0 && n >= 22 && n < 38) xreg[n - 22] = xfp[(n - 22) * 256];      // compiler-visible: it counts the wait
                    if constexpr (j == 2 && n >= 37 && n < 45) {
                        constexpr int u = n - 37;
                        if constexpr (u == 4) {
                            lds[x_nxt + tid] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                            lds[x_nxt + 256 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                        }
                        split_pair(xreg[2 * u], xreg[2 * u + 1], 1024.f, ph[u & 3], pw[u & 3]);
                        if constexpr (u == 7) {
                            lds[x_nxt + 512 + tid] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                            lds[x_nxt + 768 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            wpos += wper;
            if (wpos > wslots - wper) wpos -= wslots - wper;
            if (WINO || j == 0) {
                xpos += WINO ? 64 * 6 : 64 * 4;                    // winograd: 6 KB per wavefront and stage; direct: 16 KB per workgroup and chunk
                if (xpos > xend) xpos = xbeg;
            }
            // operands of stage s + 1 landed, workgroup barrier.  Younger VMEM that may stay in flight -- winograd: the 6
            // plane copies of stage s + 2; direct: this stage's 5 weight copies (+ the 16 window loads of the chunk)
            if constexpr (WINO) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if constexpr (j == 2) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(21) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            a3 = a3 == 2 ? 0 : a3 + 1;
            v3 = v3 == 2 ? 0 : v3 + 1;
            ++s;
        };
        if (WINO) {
            for (int k = 0; k < nstage; ++k) stage(std::integral_constant<int, 1>{});
        } else {
            for (int k = 0; k < nstage / 3; ++k) {
                stage(std::integral_constant<int, 0>{});
                stage(std::integral_constant<int, 1>{});
                stage(std::integral_constant<int, 2>{});
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const long long t2 = clock64();
        // epilogue
        // (uniform base in SGPRs + the lane as the one per-lane offset + an immediate: no per-store address registers)
        float* ow = out + (((long)blockIdx.x * tiles_per_wg + tile) * 4 + wave) * (15 * 16 * 64);     // a tile's own output
        if (WINO) {
            // the four components (one per wavefront) meet in LDS: every wavefront writes its 15 blocks (60 x 16 bytes per
            // lane), reads the other three's share of the outputs it combines (60 x 16 bytes), 2 adds per output
            // (two phases of 30 x 16 bytes per lane: the 245 KB of a workgroup's components do not fit the LDS at once)
            static_for<2>([&](auto pc) __attribute__((always_inline)) {
                constexpr int p = decltype(pc)::value;
                static_for<30>([&](auto ec) __attribute__((always_inline)) {
                    constexpr int e = p * 30 + decltype(ec)::value, blk = e / 4, q = e % 4;
                    u32x4 v;
                    v[0] = __builtin_bit_cast(unsigned, acc[blk / 3][blk % 3][4 * q]);
                    v[1] = __builtin_bit_cast(unsigned, acc[blk / 3][blk % 3][4 * q + 1]);
                    v[2] = __builtin_bit_cast(unsigned, acc[blk / 3][blk % 3][4 * q + 2]);
                    v[3] = __builtin_bit_cast(unsigned, acc[blk / 3][blk % 3][4 * q + 3]);
                    lds[(wave * 30 + decltype(ec)::value) * 64 + lane] = v;
                });
                __syncthreads();
                static_for<30>([&](auto ec) __attribute__((always_inline)) {
                    constexpr int e = p * 30 + decltype(ec)::value, blk = e / 4, q = e % 4;
                    const u32x4 v = lds[(((wave + 1 + e % 3) & 3) * 30 + decltype(ec)::value) * 64 + lane];
                    acc[blk / 3][blk % 3][4 * q] += __builtin_bit_cast(float, v[0]) - __builtin_bit_cast(float, v[1]);
                    acc[blk / 3][blk % 3][4 * q + 1] += __builtin_bit_cast(float, v[2]) + __builtin_bit_cast(float, v[3]);
                });
                __syncthreads();
            });
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int r = 0; r < 8; ++r) (ow + (i * 3 + k) * 8 * 64)[r * 64 + lane] = acc[i][k][r] + acc[i][k][r + 8];
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) (ow + (i * 3 + k) * 16 * 64)[r * 64 + lane] = acc[i][k][r];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long t3 = clock64();
        t_loop += t2 - t1;
        t_tile += t3 - t0;
    }
    if (tid == 0 && blockIdx.x == 7) { clk[0] = t_loop; clk[1] = t_tile; }
}


__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

constexpr int R_MPAD = 320;                  // rows of the packed weight planes
constexpr int R_STAGE = 2 * 2 * R_MPAD;      // 16-byte slots of one stage's slab: [plane][8-channel group][Mpad]
constexpr int R_X = 1024;                    // slots of one X window buffer
constexpr int R_TOTAL = 3 * R_X + 1;              // three X buffers (two used without FLAGS 32) + the flag slot

// FLAGS (ablation of the rega stage): 1 = no A loads, 2 = no window path, 4 = no barrier, 8 = no B reads,
// 16 = the A loads are issued but the MFMAs read a constant fragment set (issue cost without the wait for the data),
// 64 = THREE A register sets, the loads run two stages ahead (VMEM returns in order: an L2-hit A load queued behind the
//      HBM-miss window loads waits for them), 128 = the window loads in two batches of 8 (stages j = 2 of the previous chunk and
//      j = 0) right behind the A loads instead of 16 in one stage,
// 256 = the A loads every 4th slot (all wavefronts in the same slots), 512 = every 4th slot, wavefront w in slots 4 k + w (the
//      main loop is compiled once per wavefront index): the four wavefronts leave a barrier in lockstep and would otherwise
//      hand the CU's one texture-address path four 1 KB loads per 32-cycle slot, twice what it takes; the window loads fill
//      the other three slots of a group,
// 1024 (with 256 / 512) = the window split two VALU at a time in the free slots of the second tap's stage (one LDS write per
//      slot) instead of whole 4-VALU pairs + two writes in every other slot; B reads in the free slots of the last groups,
// 32 = no s_barrier in the loop: three X buffers, a wavefront raises an LDS flag behind its window writes and the readers
//      of the next chunk check the four flags (a wavefront may run a chunk ahead of its neighbours)
template <int FLAGS, int AUX = 0>
__global__ __launch_bounds__(256, 1) void probe_rega(const u32x4* __restrict__ wsrc, long wslots, const u32x4* __restrict__ xsrc,
                                                     long xslots, float* __restrict__ out, long long* __restrict__ clk,
                                                     int tiles_per_wg, int nstage, int stagger) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, nl = lane & 31, h = lane >> 5;
    f32x16 acc[5][3];
    f16x8 af[3][2][5], bf[2][2][3];                                // [register set][plane][block]; the third A set: FLAGS 64
    f16x8 afc[2][5];
    float xreg[16];
    unsigned ph[8], pw[8];
    for (int i = 0; i < 16; ++i) xreg[i] = 1.f;
    for (int i = 0; i < 10; ++i) afc[i / 5][i % 5] = __builtin_bit_cast(f16x8, wsrc[(i * 64 + lane) * 3 + 1]);
    volatile int* flags = reinterpret_cast<volatile int*>(lds + 3 * R_X);
    if (tid < 4) flags[tid] = 0;
    __syncthreads();
    int chunk_abs = 0;                                             // chunks done by this workgroup (flag values)
    long long t_loop = 0, t_tile = 0;
    const long long w_begin = wall_clock64(), c_begin = clock64();
    if (stagger > 0) {
        const int grp = (blockIdx.x >> 3) & 3;
        const long long tw = clock64() + (long long)grp * stagger;
        while (clock64() < tw) __builtin_amdgcn_s_sleep(16);
    }
    const __amdgpu_buffer_rsrc_t wr = mk_rsrc(wsrc, (unsigned)(wslots * 16));
    const int nslab = (int)(wslots / R_STAGE);                     // stages' worth of slabs in the weight buffer
    const int wvoff = (h * R_MPAD + wm * 160 + nl) * 16;           // the lane's slot inside a (plane) run pair
    const long xshare = xslots / gridDim.x;
    const long xbeg = (long)blockIdx.x * xshare + wave * (xshare / 4);
    const long xend = xbeg + xshare / 4 - 64 * 48;
    long xpos = xbeg;
    const float* xf = reinterpret_cast<const float*>(xsrc);
    int slab = 0;                                                  // slab index of the stage whose A is fetched next
    // A fragments of the stage in slab `SL_` into register set SET_, plane P_, row block MT_ (one 1 KB wave load)
#define LOAD_A(SET_, P_, MT_, SL_)                                                                            \
    af[SET_][P_][MT_] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(                       \
        wr, wvoff + (MT_) * 512, ((SL_) * 2 + (P_)) * (2 * R_MPAD * 16), AUX));
    auto body = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value;
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
        const long long t0 = clock64();
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
        // prologue: A of stage 0 into set 0, the window of chunk 0 split into X buffer 0, B of stage 0 into set 0
        static_for<10>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            LOAD_A(0, 1 - i / 5, i % 5, slab)
            if constexpr ((FLAGS & (1 | 64)) != 0) LOAD_A(1, 1 - i / 5, i % 5, slab + 1)      // ablation: both sets hold real data; 64: stage 1
        });
        slab = slab + 1 == nslab ? 0 : slab + 1;
        if constexpr ((FLAGS & 64) != 0) slab = slab + 1 == nslab ? 0 : slab + 1;
        for (int r = 0; r < 16; ++r) xreg[r] = xf[xpos * 4 + (long)r * 256 + tid];
        static_for<8>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            split_pair(xreg[2 * u], xreg[2 * u + 1], 1024.f, ph[u & 3], pw[u & 3]);
            if constexpr (u == 3) { lds[tid] = u32x4{ph[0], ph[1], ph[2], ph[3]}; lds[256 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]}; }
            if constexpr (u == 7) { lds[512 + tid] = u32x4{ph[0], ph[1], ph[2], ph[3]}; lds[768 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]}; }
        });
        __syncthreads();
        static_for<6>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            bf[0][i / 3][i % 3] = __builtin_bit_cast(f16x8, lds[i * 64 + lane]);
            if constexpr ((FLAGS & 8) != 0) bf[1][i / 3][i % 3] = __builtin_bit_cast(f16x8, lds[i * 64 + 2 + lane]);
        });
        const long long t1 = clock64();
        int s = 0;
        // stage with compile-time tap J and register-set parity PAR
        auto stage = [&](auto jc, auto pc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, par = decltype(pc)::value;
            constexpr int aset = (FLAGS & 64) ? j : par, aset_ld = (FLAGS & 64) ? (j + 2) % 3 : par ^ 1;
            const int chunk = s / 3;
            constexpr int NXB = (FLAGS & 32) ? 3 : 2;
            const int x_nxt_stage = (((j == 2 ? chunk + 1 : chunk)) % NXB) * R_X;   // X buffer of stage s + 1
            const int x_wr = ((chunk + 1) % NXB) * R_X;                            // X buffer of the next chunk
            const float* xfp = xf + xpos * 4 + tid;
            const int sl = slab;
            static_for<NM>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                constexpr int blk = n % 15, mt = blk / 3, nt = blk % 3, term = n / 15;
                constexpr int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;      // lo*hi, hi*lo, hi*hi
                if constexpr ((FLAGS & 16) != 0)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[pa][mt], bf[par][pb][nt], acc[mt][nt], 0, 0, 0);
                else
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[aset][pa][mt], bf[par][pb][nt], acc[mt][nt], 0, 0, 0);
                // A of stage s + 1 (FLAGS 64: s + 2) -> the other register set: 10 wave loads, one per slot
                constexpr bool SPREAD = (FLAGS & (256 | 512)) != 0;
                constexpr int sub = SPREAD ? (n % 4 - W + 4) % 4 : 0, grp = n / 4;
                if constexpr (!SPREAD && n < 10 && !(FLAGS & 1)) LOAD_A(aset_ld, 1 - n / 5, n % 5, sl)          // A.lo first: the first term uses it
                if constexpr (SPREAD && sub == 0 && grp < 10 && !(FLAGS & 1)) LOAD_A(aset_ld, 1 - grp / 5, grp % 5, sl)
                if constexpr (SPREAD && sub != 0 && grp * 3 + sub - 1 < 16 && !(FLAGS & 2) && j == 0) xreg[grp * 3 + sub - 1] = xfp[(grp * 3 + sub - 1) * 256];
                // the window of the next chunk: 16 loads in the first tap's stage, split + written in the second tap's
                if constexpr (!(FLAGS & (2 | 128 | 256 | 512)) && j == 0 && n >= 12 && n < 28) xreg[n - 12] = xfp[(n - 12) * 256];
                if constexpr ((FLAGS & 128) && !(FLAGS & 2) && j == 2 && n >= 10 && n < 18) xreg[n - 10] = xfp[(n - 10 + 16) * 256];
                if constexpr ((FLAGS & 128) && !(FLAGS & 2) && j == 0 && n >= 10 && n < 18) xreg[n - 2] = xfp[(n - 2) * 256];
                if constexpr ((FLAGS & 1024) && SPREAD && !(FLAGS & 2) && j == 1 && sub != 0) {
                    constexpr int f = grp * 3 + sub - 1;
                    if constexpr (f < 16) {
                        constexpr int u = f / 2;
                        if constexpr (f % 2 == 0) split_hi(xreg[2 * u], xreg[2 * u + 1], 1024.f, ph[u]);
                        else split_lo(xreg[2 * u], xreg[2 * u + 1], 1024.f, ph[u], pw[u]);
                    }
                    if constexpr (f == 9) lds[x_wr + tid] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                    if constexpr (f == 10) lds[x_wr + 256 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                    if constexpr (f == 16) lds[x_wr + 512 + tid] = u32x4{ph[4], ph[5], ph[6], ph[7]};
                    if constexpr (f == 17) lds[x_wr + 768 + tid] = u32x4{pw[4], pw[5], pw[6], pw[7]};
                }
                if constexpr (!(FLAGS & (2 | 1024)) && j == 1 && n >= 12 && n < 28 && ((n - 12) & 1) == 0) {
                    constexpr int u = (n - 12) / 2;
                    if constexpr (u == 4) {
                        lds[x_wr + tid] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                        lds[x_wr + 256 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                    }
                    split_pair(xreg[2 * u], xreg[2 * u + 1], 1024.f, ph[u & 3], pw[u & 3]);
                    if constexpr (u == 7) {
                        lds[x_wr + 512 + tid] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                        lds[x_wr + 768 + tid] = u32x4{pw[0], pw[1], pw[2], pw[3]};
                        if constexpr ((FLAGS & 32) != 0) {                       // LDS executes a wavefront's operations in order
                            asm volatile("" ::: "memory");
                            if (lane == 0) flags[wave] = chunk_abs + 1;
                        }
                    }
                }
                if constexpr ((FLAGS & 32) != 0 && j == 2 && n == 32) {          // every wavefront's share of the next window is in LDS
                    for (;;) {
                        const int f0 = flags[0], f1 = flags[1], f2 = flags[2], f3 = flags[3];
                        const int lo01 = f0 < f1 ? f0 : f1, lo23 = f2 < f3 ? f2 : f3;
                        if (__builtin_amdgcn_readfirstlane(lo01 < lo23 ? lo01 : lo23) > chunk_abs) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    asm volatile("" ::: "memory");
                }
                // B of stage s + 1 -> the other register set: 6 fragment reads
                constexpr bool BSLOT = (FLAGS & 1024) ? (sub != 0 && grp >= 8 && grp < 10) : (n >= 34 && n < 40);
                if constexpr (BSLOT && !(FLAGS & 8)) {
                    constexpr int i = (FLAGS & 1024) ? (grp - 8) * 3 + sub - 1 : n - 34;
                    bf[par ^ 1][i / 3][i % 3] = __builtin_bit_cast(f16x8, lds[x_nxt_stage + (i / 3) * 512 + (i % 3) * 32 + ((j + 1) % 3) * 2 + lane]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            slab = slab + 1 == nslab ? 0 : slab + 1;
            if (j == 0) {
                xpos += 64 * 4;
                if (xpos > xend) xpos = xbeg;
            }
            // the window of the next chunk is complete in LDS (every wavefront wrote its columns): ONE barrier per chunk
            if constexpr (j == 1 && !(FLAGS & (4 | 32))) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (j == 2) ++chunk_abs;
            ++s;
        };
        for (int k = 0; k < nstage / 6; ++k) {
            stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const long long t2 = clock64();
        float* ow = out + (((long)blockIdx.x * tiles_per_wg + tile) * 4 + wave) * (15 * 16 * 64);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) (ow + (i * 3 + k) * 16 * 64)[r * 64 + lane] = acc[i][k][r];
        if constexpr ((FLAGS & 16) != 0) {
            float keep = 0.f;
            for (int q = 0; q < 30; ++q) keep += (float)af[q / 10][(q / 5) & 1][q % 5][0];
            ow[lane] = keep;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long t3 = clock64();
        t_loop += t2 - t1;
        t_tile += t3 - t0;
    }
    };
    if constexpr ((FLAGS & 512) != 0) {
        switch (wave) {
            case 0: body(std::integral_constant<int, 0>{}); break;
            case 1: body(std::integral_constant<int, 1>{}); break;
            case 2: body(std::integral_constant<int, 2>{}); break;
            default: body(std::integral_constant<int, 3>{}); break;
        }
    } else body(std::integral_constant<int, 0>{});
#undef LOAD_A
    if (tid == 0) { clk[4 * blockIdx.x] = t_loop; clk[4 * blockIdx.x + 1] = t_tile; clk[4 * blockIdx.x + 2] = wall_clock64() - w_begin; clk[4 * blockIdx.x + 3] = clock64() - c_begin; }
}

template <int FLAGS, int AUX = 0>
static void run_rega(const u32x4* w, long wslots, const u32x4* x, long xslots, float* out, long long* clk,
                     int tiles_per_wg, int nstage, int stagger) {
    const int lds_bytes = R_TOTAL * 16;
    CK(hipFuncSetAttribute((const void*)probe_rega<FLAGS, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<long long> c(4 * 256), cb(4 * 256);
    // SUSTAINED: the clock follows the power of the last milliseconds, so a variant is run back to back for ~60 ms
    // (PROBE_REPS launches, default 400) and the second half is timed
    const char* re = getenv("PROBE_REPS");
    const int reps = re ? atoi(re) : 400;
    for (int rep = 0; rep < reps; ++rep) {
        if (rep == reps / 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe_rega<FLAGS, AUX>), dim3(256), dim3(256), lds_bytes, 0, w, wslots, x, xslots, out, clk, tiles_per_wg, nstage, stagger);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms / (reps - reps / 2); }
    CK(hipMemcpy(cb.data(), clk, 4 * 256 * 8, hipMemcpyDeviceToHost)); c = cb;
    // over the 256 workgroups: mean / max cycles per tile, mean stage cycles, mean shader clock (cycles / 100 MHz wall ticks)
    double tile = 0, tmax = 0, loop = 0, mhz = 0, wall = 0;
    for (int i = 0; i < 256; ++i) {
        tile += (double)c[4 * i + 1] / tiles_per_wg; loop += (double)c[4 * i] / tiles_per_wg;
        if ((double)c[4 * i + 1] / tiles_per_wg > tmax) tmax = (double)c[4 * i + 1] / tiles_per_wg;
        mhz += (double)c[4 * i + 3] / ((double)c[4 * i + 2] / 100.0); wall += (double)c[4 * i + 2] / 100.0;
    }
    tile /= 256; loop /= 256; mhz /= 256; wall /= 256;
    printf("rega f=%-2d aux=%d s=%-5d: %7.1f us per launch (in-kernel wall %6.1f us, %4.0f MHz); per tile %7.0f cycles (max %7.0f), main loop "
           "%7.0f = %5.0f per stage (45 MFMAs = 1440)\n", FLAGS, AUX, stagger, best * 1e3, wall, mhz, tile, tmax, loop, loop / nstage);
}

template <bool WINO>
static void run(const char* name, const u32x4* w, long wslots, const u32x4* x, long xslots, float* out,
                long long* clk, int tiles_per_wg, int nstage) {
    const int lds_bytes = (WINO ? W_TOTAL : D_TOTAL) * 16;
    CK(hipFuncSetAttribute((const void*)probe<WINO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    printf("%s: launching with %d bytes of LDS\n", name, lds_bytes); fflush(stdout);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    long long c[2] = {0, 0};
    const char* re = getenv("PROBE_REPS");
    const int reps = re ? atoi(re) : 400;
    for (int rep = 0; rep < reps; ++rep) {
        if (rep == reps / 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<WINO>, dim3(256), dim3(256), lds_bytes, 0, w, wslots, x, xslots, out, clk, tiles_per_wg, nstage);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms / (reps - reps / 2); }
    CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
    printf("%-9s %2d tiles per CU x %2d stages: %8.1f us per launch; per tile %7.0f cycles, main loop %7.0f = %6.0f per stage "
           "(45 MFMAs = 1440)\n", name, tiles_per_wg, nstage, best * 1e3, (double)c[1] / tiles_per_wg,
           (double)c[0] / tiles_per_wg, (double)c[0] / tiles_per_wg / nstage);
}

int main() {
    const long wslots = (2L << 20) / 16;               // 2 MB of "weights": L2-resident
    const long xslots = (1024L << 20) / 16;            // 1 GB of "activations": streamed from HBM
    std::vector<unsigned short> h(8 << 20);
    srand(1);
    for (auto& v : h) {                                // random f16 in [-2, 2): sign, exponent 12..15, random mantissa
        v = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 0x3ff));
    }
    u32x4 *w, *x; float* out; long long* clk;
    CK(hipMalloc(&w, wslots * 16)); CK(hipMalloc(&x, xslots * 16)); CK(hipMalloc(&out, 256L * 4 * 4 * 15 * 16 * 64 * 4)); CK(hipMalloc(&clk, 4 * 256 * 8));
    CK(hipMemcpy(w, h.data(), wslots * 16, hipMemcpyHostToDevice));
    for (long off = 0; off < xslots * 16; off += (long)h.size() * 2)
        CK(hipMemcpy((char*)x + off, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    // one conv-sized launch at B = 256: 512 tiles of 320 x 192 / 1 024 half tiles of 160 x 192 on 256 CUs
    const char* only = getenv("PROBE_ONLY");
    for (int rep = 0; rep < 2; ++rep) {
        if (!only || only[0] == 'd') run<false>("direct", w, wslots, x, xslots, out, clk, 2, 60);
        if (!only || only[0] == 'w') run<true>("winograd", w, wslots, x, xslots, out, clk, 4, 20);
        if (!only || only[0] == 'r' || only[0] == 's' || only[0] == 'e' || only[0] == 'c') {
          if (only && only[0] == 'c') {          // cache policy of the A loads (aux: 1 = sc0, 2 = nt, 3 = both, 16 = sc1), sustained
            run_rega<1280, 0>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280, 1>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280, 2>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280, 3>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280, 16>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280, 0>(w, wslots, x, xslots, out, clk, 2, 60, 0);
          } else if (only && only[0] == 'e') {          // where the wall time (= energy, the chip is power-limited) of the kept form goes
            run_rega<1280>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1281>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1282>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1288>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1284>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1283>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<15>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280>(w, wslots, x, xslots, out, clk, 2, 60, 0);
          } else if (only && only[0] == 's') {          // the short list: today's structure against the candidates, sustained
            run<false>("direct", w, wslots, x, xslots, out, clk, 2, 60);
            run_rega<0>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<256>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1536>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<15>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run<false>("direct", w, wslots, x, xslots, out, clk, 2, 60);
          } else {
            run_rega<0>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<16>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<256>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1280>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1536>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<258>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<6>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<1>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<2>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<4>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<8>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<3>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<15>(w, wslots, x, xslots, out, clk, 2, 60, 0);
            run_rega<0>(w, wslots, x, xslots, out, clk, 2, 60, 7000);
          }
        }
    }
    return 0;
}
