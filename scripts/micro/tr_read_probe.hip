// Empirical semantics of gfx950's ds_read_b64_tr_b16 (no ISA text in this container): LDS is filled with
// lds16[i] = i (16-bit), every lane issues one transpose read at a chosen byte address and dumps the four 16-bit
// values it received.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const int* __restrict__ addr, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds16[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds16[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)addr[threadIdx.x] + (unsigned)(size_t)(&lds16[0]);      // LDS byte address
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(v[0] & 0xffffu);
    out[threadIdx.x * 4 + 1] = (unsigned short)(v[0] >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v[1] & 0xffffu);
    out[threadIdx.x * 4 + 3] = (unsigned short)(v[1] >> 16);
}

static void run(const char* title, const std::vector<int>& addr) {
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * sizeof(int)); hipMalloc(&d_out, 256 * sizeof(unsigned short));
    hipMemcpy(d_addr, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    std::vector<unsigned short> out(256);
    hipMemcpy(out.data(), d_out, 256 * sizeof(unsigned short), hipMemcpyDeviceToHost);
    printf("== %s  (per lane: byte address -> the four 16-bit ELEMENT INDICES received)\n", title);
    for (int l = 0; l < 64; ++l)
        printf("lane %2d addr %4d -> %4d %4d %4d %4d%s", l, addr[l], out[4 * l], out[4 * l + 1], out[4 * l + 2], out[4 * l + 3],
               (l % 2 == 1) ? "\n" : "   |   ");
    hipFree(d_addr); hipFree(d_out);
}

int main() {
    std::vector<int> a(64);
    // 1. lane l reads at l * 8 bytes (4 consecutive elements each: a plain b64 read would return 4l .. 4l+3)
    for (int l = 0; l < 64; ++l) a[l] = l * 8;
    run("addr = lane * 8", a);
    // 2. a [16 rows][row stride 64 B] image: lane l -> row (l & 15), 8-byte column (l >> 4)
    for (int l = 0; l < 64; ++l) a[l] = (l & 15) * 64 + (l >> 4) * 8;
    run("addr = (lane & 15) * 64 + (lane >> 4) * 8", a);
    // 3. rows of 32 B: lane l -> row (l & 15), 8-byte column (l >> 4)
    for (int l = 0; l < 64; ++l) a[l] = (l & 15) * 32 + (l >> 4) * 8;
    run("addr = (lane & 15) * 32 + (lane >> 4) * 8", a);
    // 4. [4 rows][stride 128 B]: lane l -> row (l & 3), 8-byte column (l >> 2)
    for (int l = 0; l < 64; ++l) a[l] = (l & 3) * 128 + (l >> 2) * 8;
    run("addr = (lane & 3) * 128 + (lane >> 2) * 8", a);
    return 0;
}
