// How fast does HBM deliver the access pattern of a split-K contraction over row-major [rows][K] operands?
// Every workgroup walks its own K range; per step it reads RUN consecutive 128-byte lines from each of `rows` rows
// that lie `stride` bytes apart (the ClipLoss score contraction: rows = 256 estimates + 128 candidates, stride =
// 4 K bytes = 172 800 (F = 120) or 1 474 560 (F = 1 024), RUN = 1 = one 32-sample stage).  RUN > 1 = longer
// contiguous runs per row and step, same bytes in total.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/strided_line_probe.hip -o /tmp/slp && /tmp/slp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int RUN, int ROWS>
__global__ __launch_bounds__(256) void walk(const char* __restrict__ base, long stride, int lines_per_wg, unsigned* out) {
    // thread -> (row32 = tid >> 3, 16-byte piece tid & 7); passes over ROWS / 32 row groups
    const int tid = threadIdx.x;
    const long l0 = (long)blockIdx.x * lines_per_wg;
    unsigned acc = 0;
    for (int l = 0; l < lines_per_wg; l += RUN) {
        u32x4 v[ROWS / 32][RUN];
#pragma unroll
        for (int g = 0; g < ROWS / 32; ++g)
#pragma unroll
            for (int r = 0; r < RUN; ++r)
                v[g][r] = *reinterpret_cast<const u32x4*>(base + (long)(g * 32 + (tid >> 3)) * stride + (l0 + l + r) * 128 + (tid & 7) * 16);
#pragma unroll
        for (int g = 0; g < ROWS / 32; ++g)
#pragma unroll
            for (int r = 0; r < RUN; ++r) acc ^= v[g][r][0] ^ v[g][r][1] ^ v[g][r][2] ^ v[g][r][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int RUN, int ROWS>
static void run(const char* d, long stride, long lines_per_row, unsigned* d_out, const char* tag) {
    const int nwg = 256;
    int lines_per_wg = (int)(lines_per_row / nwg) / 4 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((walk<RUN, ROWS>), dim3(nwg), dim3(256), 0, 0, d, stride, lines_per_wg, d_out);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<RUN, ROWS>), dim3(nwg), dim3(256), 0, 0, d, stride, lines_per_wg, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nwg * lines_per_wg * 128.0 * ROWS;
    printf("%-28s rows %3d run %d x 128 B: %7.1f us  %5.2f TB/s\n", tag, ROWS, RUN, ms / reps * 1e3, bytes * reps / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned* d_out;
    hipMalloc(&d_out, 64);
    const long strides[2] = {172800, 1474560};
    for (int si = 0; si < 2; ++si) {
        const long stride = strides[si];
        char* d;
        const size_t bytes = (size_t)stride * 384;
        hipMalloc(&d, bytes);
        hipMemset(d, 1, bytes);
        const long lines_per_row = stride / 128;
        char tag[64];
        snprintf(tag, sizeof tag, "stride %ld", stride);
        run<1, 384>(d, stride, lines_per_row, d_out, tag);
        run<2, 384>(d, stride, lines_per_row, d_out, tag);
        run<4, 384>(d, stride, lines_per_row, d_out, tag);
        run<1, 128>(d, stride, lines_per_row, d_out, tag);
        run<4, 128>(d, stride, lines_per_row, d_out, tag);
        run<8, 128>(d, stride, lines_per_row, d_out, tag);
        hipFree(d);
    }
    return 0;
}
