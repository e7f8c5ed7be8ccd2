// Probe: does an out-of-range lane of buffer_load ... lds (LDS-DMA through a buffer descriptor) write ZERO to its
// LDS slot, or leave the slot untouched?  hipcc --offload-arch=gfx950 tools/probes/lds_dma_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, int bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    const unsigned off = (lane & 1) ? 0xfffffff0u : lane * 16;   // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    float *d, *o;
    hipMalloc(&d, 1024);
    hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024, o);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("lane0 (valid): %g %g %g %g\n", r[0], r[1], r[2], r[3]);
    printf("lane1 (OOB)  : %g %g %g %g\n", r[4], r[5], r[6], r[7]);
    printf("lane2 (valid): %g %g %g %g\n", r[8], r[9], r[10], r[11]);
    printf("lane3 (OOB)  : %g %g %g %g\n", r[12], r[13], r[14], r[15]);
    return 0;
}
