// Diagnostic: what the fp32 matrix pipe delivers IN SITU, to tell a clock / power limit from stalls in the conv kernel's K loop.
// Four loop bodies with the conv kernel's per-wave MFMA stream (two 32x32 accumulators, 8 MFMAs per k-group of 8, 32 per K chunk),
// run on the conv kernel's geometry (256-thread blocks, 1..3 blocks per CU through the dynamic-LDS size):
//   0  operands from registers only (no LDS, no barrier): the matrix pipe's ceiling at the clock the chip holds under this load
//   1  + the conv loop's fragment reads (3 ds_read_b128 per k-group from a swizzled 128x32 / 64x32 image), no barrier
//   2  + one s_barrier per K chunk (32 MFMAs)
//   3  + the LDS-DMA fill of the next chunk (global_load_lds_dwordx4, 6 pieces per thread per chunk, L2-resident source)
// Prints TFLOP/s from hipEvents and the in-kernel clock (s_memtime / s_memrealtime, median over blocks).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o tools/bin/mfma_peak_probe && tools/bin/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][128][32]
    float* Bs = smem + 2 * 128 * 32;  // [2][64][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 192 * 32; i += 256) smem[i] = src[(i * 7 + blockIdx.x) & 0xfffff];
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int arow = wm * 64 + (lane & 31), brow = wn * 32 + (lane & 31), kh = lane >> 5;
    f32x4 ra[2], rb;
    ra[0] = *(const f32x4*)(As + arow * 32 + kh * 4);
    ra[1] = *(const f32x4*)(As + (arow + 32) * 32 + kh * 4);
    rb = *(const f32x4*)(Bs + brow * 32 + kh * 4);
    const int lrow = tid >> 3, slot = tid & 7;
    const float* gp = src + ((size_t)(blockIdx.x % 512) * 192 + lrow) * 32 + slot * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        // MODE 3: the conv kernel's form (64-bit per-lane addresses, M0 from a per-lane LDS pointer).  4: wave index made uniform.  5: scalar base +
        // 32-bit per-lane offset (saddr form).  6: buffer_load ... lds (resource + 32-bit offset).  7: mode 4 with half the pieces.
        // 8: mode 4 with the six pieces spread behind the MFMA groups instead of bunched at the head of the chunk.
        const int wv = (MODE >= 4) ? __builtin_amdgcn_readfirstlane(wave) : wave;
        float* la = As + (buf ^ 1) * 128 * 32 + wv * 8 * 32;
        float* lb = Bs + (buf ^ 1) * 64 * 32 + wv * 8 * 32;
        const float* gsrc = gp + (size_t)(c & 15) * 6144;
        auto piece = [&](int j) {
            float* l = j < 4 ? la + j * 32 * 32 : lb + (j - 4) * 32 * 32;
            if (MODE == 5) {
                const unsigned voff = (unsigned)(((size_t)lrow * 32 + slot * 4 + (size_t)j * 32 * 32) * 4);
                const float* sb = src + (size_t)(blockIdx.x % 512) * 192 * 32 + (size_t)(c & 15) * 6144;
                const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)l;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sb), "s"(m0v) : "memory");
            } else if (MODE == 6) {
                const unsigned voff = (unsigned)(((size_t)lrow * 32 + slot * 4 + (size_t)j * 32 * 32) * 4);
                const float* sb = src + (size_t)(blockIdx.x % 512) * 192 * 32 + (size_t)(c & 15) * 6144;
                auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)sb, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
            } else {
                glds16(gsrc + (size_t)j * 32 * 32, l);
            }
        };
        if (MODE >= 3 && MODE != 8) {
#pragma unroll
            for (int j = 0; j < (MODE == 7 ? 3 : 6); ++j) piece(j);
        }
        const float* a = As + (MODE >= 3 ? buf : 0) * 128 * 32;
        const float* b = Bs + (MODE >= 3 ? buf : 0) * 64 * 32;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 af[2], bf;
            if (MODE >= 1) {
                const int q = ks * 2 + kh;
                af[0] = *(const f32x4*)(a + arow * 32 + ((q ^ ((arow >> 1) & 7)) << 2));
                af[1] = *(const f32x4*)(a + (arow + 32) * 32 + ((q ^ (((arow + 32) >> 1) & 7)) << 2));
                bf = *(const f32x4*)(b + brow * 32 + ((q ^ ((brow >> 1) & 7)) << 2));
            } else {
                af[0] = ra[0]; af[1] = ra[1]; bf = rb;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], acc[i], 0, 0, 0);
                if (MODE == 8 && ks < 3 && (e & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(ks * 2 + (e >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// MODE 0 of the probe plus F dependent-free v_fma_f32 fillers behind every MFMA: does the fp32 matrix pipe run beside the same wave's
// vector instructions (it executes at the fp32 VECTOR rate), and does a second wave per SIMD change what fillers cost?
template <int F>
__global__ __launch_bounds__(256, 2) void probe_fill(const float* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = src[tid], b0 = src[tid + 256];
    float f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = src[tid + 512 + k * 256];
    if (tid == 0) smem[0] = a0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < F; ++k) f[k & 15] = __builtin_fmaf(f[k & 15], 1.0001f, 0.25f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += f[k];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int F>
static void run_fill(int blocks_per_cu, int nchunks, const float* src, float* out, unsigned long long* stamps) {
    const int lds = (160 * 1024 / blocks_per_cu) - 1024;
    CK(hipFuncSetAttribute((const void*)probe_fill<F>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = 256 * blocks_per_cu * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe_fill<F>, dim3(grid), dim3(256), lds, 0, src, out, stamps, nchunks);
    CK(hipDeviceSynchronize());
    const int R = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(probe_fill<F>, dim3(grid), dim3(256), lds, 0, src, out, stamps, nchunks);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= R;
    const double flops = (double)grid * 4 * nchunks * 32.0 * (32.0 * 32 * 2 * 2);
    printf("fillers per MFMA %2d   blocks/CU %d (waves/SIMD %d)  %8.3f ms  %7.1f TF/s  (%.3f of 157.3)\n", F, blocks_per_cu, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    fflush(stdout);
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, int nchunks, const float* src, float* out, unsigned long long* stamps) {
    // LDS per block decides residency: 160 KiB / blocks_per_cu (minus a little), at least the 48 KiB the loop uses
    const int lds = std::max(2 * 192 * 32 * 4, (160 * 1024 / blocks_per_cu) - 1024 - (blocks_per_cu == 1 ? 0 : 0));
    CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = 256 * blocks_per_cu * 4;      // four full rounds
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds, 0, src, out, stamps, nchunks);
    CK(hipDeviceSynchronize());
    const int R = 20;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds, 0, src, out, stamps, nchunks);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= R;
    std::vector<unsigned long long> st(2 * grid);
    CK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < grid; ++b) if (st[2 * b + 1]) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double flops = (double)grid * 4 /*waves*/ * nchunks * 32.0 * (32.0 * 32 * 2 * 2);
    printf("%-44s blocks/CU %d  %8.3f ms  %7.1f TF/s  (%.3f of 157.3)  clock %.0f MHz (median; %.0f..%.0f)\n", name, blocks_per_cu, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3, clk.empty() ? 0.0 : clk[clk.size() / 2], clk.empty() ? 0.0 : clk.front(), clk.empty() ? 0.0 : clk.back());
    fflush(stdout);
}

int main() {
    float* src; float* out; unsigned long long* stamps;
    const size_t n = 1 << 22;
    CK(hipMalloc(&src, n * 4 + 65536 * 4)); CK(hipMalloc(&out, (size_t)4096 * 256 * 4)); CK(hipMalloc(&stamps, 8192 * 16));
    std::vector<float> h(n + 65536);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int nchunks = 216;     // 4 x the 54 chunks of conv112's K = 27 * 64
    if (getenv("PROBE_FILL")) {
        for (int bpc = 1; bpc <= 2; ++bpc) {
            run_fill<0>(bpc, nchunks, src, out, stamps);
            run_fill<2>(bpc, nchunks, src, out, stamps);
            run_fill<4>(bpc, nchunks, src, out, stamps);
            run_fill<8>(bpc, nchunks, src, out, stamps);
            run_fill<12>(bpc, nchunks, src, out, stamps);
            run_fill<16>(bpc, nchunks, src, out, stamps);
        }
        return 0;
    }
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0>("0 registers only", bpc, nchunks, src, out, stamps);
        run<1>("1 + fragment reads (ds_read_b128)", bpc, nchunks, src, out, stamps);
        run<2>("2 + barrier per chunk", bpc, nchunks, src, out, stamps);
        run<3>("3 + LDS-DMA fill of the next chunk", bpc, nchunks, src, out, stamps);
        if (getenv("PROBE_DMA")) {
            run<4>("4 DMA, uniform wave index", bpc, nchunks, src, out, stamps);
            run<5>("5 DMA, scalar base + 32-bit lane offset", bpc, nchunks, src, out, stamps);
            run<6>("6 DMA, buffer_load ... lds", bpc, nchunks, src, out, stamps);
            run<7>("7 DMA, half the pieces", bpc, nchunks, src, out, stamps);
            run<8>("8 DMA, pieces spread behind MFMA groups", bpc, nchunks, src, out, stamps);
        }
    }
    return 0;
}
