// Diagnostic: does buffer_load_dwordx4 ... lds (LDS-DMA through a raw buffer resource) write ZEROS for lanes whose offset is out of range?
// (If so the gather kernels need no zero line and no 64-bit address select for padding taps.)
//   hipcc --offload-arch=gfx950 -O3 tools/dma_oob_probe.hip -o /tmp/dma_oob_probe && /tmp/dma_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k(const float* src, unsigned nrec, float* out, long long delta) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + delta), 0, nrec, 0x00020000);
    // piece 0: lanes 0..31 in range, 32..63 offset 0xffffffff; piece 1: all in range, through soffset
    const unsigned voff = lane < 32 ? (unsigned)(lane * 16 - delta * 4) : 0xffffffffu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    const unsigned voff2 = (unsigned)(lane * 16 - delta * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff2, 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

int main() {
    float* src; float* out;
    CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 4096));
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i + 1);
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (long long delta : {0ll, -64ll}) {      // delta < 0: the resource base sits BELOW the allocation (a tap offset folded into the base)
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src + 1024, 0x7fffffffu, out, delta);
        CK(hipDeviceSynchronize());
        std::vector<float> o(512);
        CK(hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < 4; ++q) {
                const float want0 = l < 32 ? (float)(1024 + l * 4 + q + 1) : 0.f;
                const float want1 = (float)(1024 + 256 + l * 4 + q + 1);
                if (o[l * 4 + q] != want0) { if (bad < 4) printf("piece0 lane %d q %d: got %g want %g\n", l, q, o[l * 4 + q], want0); ++bad; }
                if (o[256 + l * 4 + q] != want1) { if (bad < 4) printf("piece1 lane %d q %d: got %g want %g\n", l, q, o[256 + l * 4 + q], want1); ++bad; }
            }
        printf("delta %lld: %s (%d mismatches)\n", delta, bad ? "FAIL" : "ok: out-of-range lanes wrote zeros, soffset added", bad);
    }
    return 0;
}
