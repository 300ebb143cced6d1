// Does a STRUCTURED buffer descriptor (stride = row bytes, idxen) zero-fill reads whose byte offset falls outside
// [0, stride) ?  (Would give per-row zero padding of a [rows][T] tensor for free.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* p, int rows, int T, int* idx, int* off, unsigned* out, int n, int mode) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = (int)(unsigned)u;
    d[1] = (int)((unsigned)(u >> 32) & 0xffffu) | ((T * 4) << 16);      // stride in bits 48..61
    d[2] = rows;                                                         // num_records = rows (structured)
    d[3] = mode;
    const int i = threadIdx.x;
    if (i >= n) return;
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v) : "v"(((unsigned long long)(unsigned)off[i] << 32) | (unsigned)idx[i]), "s"(d) : "memory");
    for (int k = 0; k < 4; ++k) out[i * 4 + k] = v[k];
}

int main() {
    const int rows = 4, T = 10;
    std::vector<float> h(rows * T + 64);
    for (int i = 0; i < (int)h.size(); ++i) h[i] = 100.f + i;
    float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // (row, byte offset): in range, straddling the row end, past the row end, negative
    int hi[] = {1, 1, 1, 1, 0, 3, 3, 4};
    int ho[] = {0, 7 * 4, 10 * 4, -2 * 4, -4 * 4, 8 * 4, 12 * 4, 0};
    const int n = 8;
    int *di, *dof; unsigned* dout;
    hipMalloc(&di, sizeof(hi)); hipMalloc(&dof, sizeof(ho)); hipMalloc(&dout, n * 16);
    hipMemcpy(di, hi, sizeof(hi), hipMemcpyHostToDevice); hipMemcpy(dof, ho, sizeof(ho), hipMemcpyHostToDevice);
    for (int mode : {0x00020000, 0x00024000 /* + bit 14? */, 0x10020000}) {
        hipMemset(dout, 0xff, n * 16);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, rows, T, di, dof, dout, n, mode);
        std::vector<float> o(n * 4); hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
        printf("word3 = 0x%08x\n", mode);
        for (int i = 0; i < n; ++i)
            printf("  row %d off %3d B -> %g %g %g %g   (in-range values would be %g..)\n", hi[i], ho[i], o[i*4], o[i*4+1], o[i*4+2], o[i*4+3],
                   100.f + hi[i] * T + ho[i] / 4);
    }
    return 0;
}
