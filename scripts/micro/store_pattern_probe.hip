// Is the conv epilogue's output store bound by HBM write bandwidth or by store ISSUE?  512 workgroups (2 per CU, 4
// wavefronts each) write a [256 segments][320 rows][360 columns] fp32 tensor tile by tile (320 x 192 per workgroup,
// 160 x 96 per wavefront as 5 x 3 blocks of 32 x 32), nothing else:
//   mode 0: the MFMA C/D layout as the conv epilogue stores it today -- per block 16 dword stores per lane (lane =
//           column, 4 consecutive rows per group), 240 per lane;
//   mode 1: row-contiguous 16-byte stores (what an LDS transpose of each block would allow), 60 per lane.
// Same bytes, same tiles.   hipcc --offload-arch=gfx950 -O2 scripts/micro/store_pattern_probe.hip -o /tmp/spp && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int B = 256, M = 320, T = 360;

template <int MODE>
__global__ __launch_bounds__(256, 1) void store_tiles(float* __restrict__ y, float seed) {
    const int b = blockIdx.x >> 1, ntile = blockIdx.x & 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, nl = lane & 31, h = lane >> 5;
    float* yb = y + (long)b * M * T;
    const int n0 = ntile * 192 + wn * 96;
    float v = seed + lane;                                 // the values do not matter
    if (MODE == 0) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = n0 + nt * 32 + nl;
            if (col < T) {
#pragma unroll
                for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * 160 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        yb[(long)row * T + col] = v + r;
                    }
            }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = n0 + nt * 32 + (lane & 7) * 4;
            if (col < T) {
#pragma unroll
                for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int row = wm * 160 + mt * 32 + p * 8 + (lane >> 3);
                        *reinterpret_cast<float4*>(yb + (long)row * T + col) = float4{v, v + 1, v + 2, v + p};
                    }
            }
        }
    }
}

template <int MODE>
static void run(float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_tiles<MODE>, dim3(512), dim3(256), 0, 0, d, 1.f);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_tiles<MODE>, dim3(512), dim3(256), 0, 0, d, (float)i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)B * M * T * 4;
    printf("mode %d (%s): %7.1f us per tensor, %5.2f TB/s\n", MODE, MODE ? "60 x 16-byte row-contiguous stores per lane" : "240 dword stores per lane, C/D layout",
           ms / reps * 1e3, bytes * reps / (ms * 1e-3) / 1e12);
}

int main() {
    float* d;
    hipMalloc(&d, (size_t)B * M * T * 4);
    run<0>(d); run<1>(d); run<0>(d); run<1>(d);
    return 0;
}
