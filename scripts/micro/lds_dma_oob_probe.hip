// Does an LDS-DMA load (buffer_load_dwordx4 ... lds) whose per-lane offset is OUT OF RANGE of the buffer descriptor
// write zeros into its LDS slot, or leave the slot untouched?  (Decides whether the conv's input window -- whose
// out-of-segment columns are the conv's zero padding -- can be staged by DMA from pre-split f16 planes.)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_dma_oob_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const unsigned* __restrict__ src, int nbytes, unsigned* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    const unsigned long long u = (unsigned long long)src;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, nbytes, 0x00020000);
    // lanes 0..47 in range (16 bytes each = 768 bytes), lanes 48..63 past the 768-byte buffer; lane 5 far out of range
    int voff = threadIdx.x * 16;
    if (threadIdx.x == 5) voff = 0x40000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = lds[i];
}

int main() {
    std::vector<unsigned> h(64 * 4);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    unsigned *d_src, *d_out;
    hipMalloc(&d_src, 1024); hipMalloc(&d_out, 1024);
    hipMemcpy(d_src, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, 768, d_out);
    std::vector<unsigned> o(256);
    hipError_t e = hipMemcpy(o.data(), d_out, 1024, hipMemcpyDeviceToHost);
    printf("status %d; LDS slot (lane) -> first dword after the DMA (source value 0x1000 + 4 * lane; 0xAAAAAAAA = untouched)\n", (int)e);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %08x%s", l, o[4 * l], (l % 4 == 3) ? "\n" : "   ");
    return 0;
}
