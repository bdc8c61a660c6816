// Feasibility probe: fp32 GEMM C[M][N] = A[M][K] * W[N][K]^T computed on the bf16 matrix cores with
// the 3-way exact split a = h + m + l (8+8+8 significant bits) and the six products whose weight is
// >= 2^-16 (hh, hm, mh, hl, lh, mm), fp32 accumulate.  Prints TFLOP/s (algorithmic 2MNK) and the
// error against an fp64 reference next to a plain fp32 fmaf-chain's error.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_x6_probe.hip -o gpurun_out/x6probe && gpurun_out/x6probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {      // RNE, a in the low half
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// split 8 consecutive floats into three planes of 8 bf16 (16 B each)
__device__ __forceinline__ void split8(const float* v, u32x4& H, u32x4& M, u32x4& L) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a0 = v[2 * q], a1 = v[2 * q + 1];
        const uint32_t h = pk_bf16(a0, a1);
        const float r0 = a0 - lo_f(h), r1 = a1 - hi_f(h);
        const uint32_t m = pk_bf16(r0, r1);
        const float s0 = r0 - lo_f(m), s1 = r1 - hi_f(m);
        const uint32_t l = pk_bf16(s0, s1);
        H[q] = h; M[q] = m; L[q] = l;
    }
}

__global__ void split_planes(const float* __restrict__ w, uint16_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = w[i];
    const uint32_t h = pk_bf16(a, 0.f) & 0xffff;
    const float r = a - __uint_as_float(h << 16);
    const uint32_t m = pk_bf16(r, 0.f) & 0xffff;
    const float s = r - __uint_as_float(m << 16);
    const uint32_t l = pk_bf16(s, 0.f) & 0xffff;
    out[i] = h; out[n + i] = m; out[2 * n + i] = l;
}

// LDS tile image: [plane][row][BK bf16]; 16-B chunks XOR-swizzled so that the 16 rows a quarter-wave
// reads with ds_read_b128 (same chunk, consecutive rows) land in 16 different 4-bank groups.
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    if (BK == 32) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
    return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4);
}

// Block tile BM x BN, NT threads = (BM/64) x (BN/WN) waves, wave tile 64 x WN.
template <int TERMS, int BM, int BN, int WN, int BK, int NBUF, int SCHED>
__global__ __launch_bounds__((BM / 64) * (BN / WN) * 64, 1) void gemm_x6(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Wp, int64_t wplane,
                                                      float* __restrict__ C, int ldc, int M, int N, int K) {
    constexpr int NT = (BM / 64) * (BN / WN) * 64;
    constexpr int APL = BM * BK * 2, BPL = BN * BK * 2;      // bytes per plane
    constexpr int STAGE = 3 * (APL + BPL);
    constexpr int CPR = BK / 8;                    // 16-B chunks per row
    constexpr int RPP = NT / CPR;                  // rows staged per pass
    constexpr int NJA = BM / RPP, NJB = BN / RPP;
    constexpr int TN = WN / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w / (BN / WN), wn = w % (BN / WN);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;      // N-tiles of one M-tile adjacent: A comes from L2
    const int srow = t / CPR, sc = t % CPR;

    float raa[1][NJA][8];
    u32x4 rbb[1][3][NJB];
    auto gload = [&](int k0, auto setc) {
        constexpr int S = decltype(setc)::value;
        auto& ra = raa[S];
        auto& rb = rbb[S];
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int row = m0 + srow + RPP * j;
            const float4* p = (const float4*)(A + (int64_t)(row < M ? row : M - 1) * lda + k0 + sc * 8);
            const float4 x = p[0], y = p[1];
            ra[j][0] = x.x; ra[j][1] = x.y; ra[j][2] = x.z; ra[j][3] = x.w;
            ra[j][4] = y.x; ra[j][5] = y.y; ra[j][6] = y.z; ra[j][7] = y.w;
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j)
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3)
                rb[p3][j] = *(const u32x4*)(Wp + p3 * wplane + (int64_t)(n0 + srow + RPP * j) * K + k0 + sc * 8);
    };
    auto lstore = [&](uint8_t* base, auto setc) {
        constexpr int S = decltype(setc)::value;
        auto& ra = raa[S];
        auto& rb = rbb[S];
        uint8_t* As = base;
        uint8_t* Bs = base + 3 * APL;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            u32x4 H, Mi, L;
            split8(ra[j], H, Mi, L);
            const int o = lds_off<BK>(srow + RPP * j, sc);
            *(u32x4*)(As + o) = H;
            *(u32x4*)(As + APL + o) = Mi;
            *(u32x4*)(As + 2 * APL + o) = L;
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int o = lds_off<BK>(srow + RPP * j, sc);
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3) *(u32x4*)(Bs + p3 * BPL + o) = rb[p3][j];
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int r32 = lane & 31, h = lane >> 5;
    auto compute = [&](const uint8_t* base, int s) {
        const uint8_t* As = base;
        const uint8_t* Bs = base + 3 * APL;
        bf16x8 a[2][3], b[TN][3];
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][p3] = *(const bf16x8*)(As + p3 * APL + lds_off<BK>(wm * 64 + i * 32 + r32, 2 * s + h));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j][p3] = *(const bf16x8*)(Bs + p3 * BPL + lds_off<BK>(wn * WN + j * 32 + r32, 2 * s + h));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                if (TERMS >= 6) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                }
                if (TERMS >= 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nch = K / BK;
    if (NBUF == 2) {
        // LDS[i&1] = chunk i; registers hold chunk i+1 (loaded during iteration i, stored to LDS[(i&1)^1] mid-iteration)
        gload(0, S0{});
        lstore(lds, S0{});
        __syncthreads();
        auto body = [&](int i, auto dostore) {
            constexpr bool ST = decltype(dostore)::value;
            uint8_t* bc = lds + (i & 1) * STAGE;
            uint8_t* bn = lds + ((i & 1) ^ 1) * STAGE;
            if (ST) gload((i + 1) * BK, S0{});
            compute(bc, 0);
            if (BK == 32) {
                if (ST) lstore(bn, S0{});
                compute(bc, 1);
            } else if (ST) {
                lstore(bn, S0{});
            }
            if (SCHED && BK == 32 && TERMS == 6 && ST) {
                // s=0 fragment reads, then each s=0 MFMA carries 4 split-VALU ops (and the LDS writes) in its shadow
                __builtin_amdgcn_sched_group_barrier(0x020, NJA * 2 + NJB * 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 6 + 3 * TN, 0);
#pragma unroll
                for (int q = 0; q < 12 * TN; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if (q % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 6 + 3 * TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12 * TN, 0);
            }
            __syncthreads();
        };
        int i = 0;
        for (; i + 1 < nch; ++i) body(i, std::true_type{});      // branch-free body = one scheduling region
        body(i, std::false_type{});
    } else {
        gload(0, S0{});
        lstore(lds, S0{});
        __syncthreads();
        for (int k0 = 0; k0 < K; k0 += BK) {
            const bool more = k0 + BK < K;
            if (more) gload(k0 + BK, S0{});
            compute(lds, 0);
            if (BK == 32) compute(lds, 1);
            __syncthreads();
            if (more) {
                lstore(lds, S0{});
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = n0 + wn * WN + j * 32 + r32;
                if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[i][j][r];
            }
}

template <int TERMS, int BM, int BN, int WN, int BK, int NBUF, int SCHED = 0>
static void run(const char* tag, const float* dA, const uint16_t* dWp, float* dC, int M, int N, int K, std::vector<float>& hC,
                const std::vector<float>& hA, const std::vector<float>& hW) {
    if (N % BN) { printf("%-22s skipped (N %% %d)\n", tag, BN); return; }
    constexpr int NT = (BM / 64) * (BN / WN) * 64;
    const int64_t nW = (int64_t)N * K;
    dim3 grid(N / BN, (M + BM - 1) / BM);
    const int shm = NBUF * 3 * (BM + BN) * BK * 2;
    CK(hipFuncSetAttribute((const void*)gemm_x6<TERMS, BM, BN, WN, BK, NBUF, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() { gemm_x6<TERMS, BM, BN, WN, BK, NBUF, SCHED><<<grid, NT, shm>>>(dA, K, dWp, nW, dC, N, M, N, K); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double e_emul = 0, e_f32 = 0, scale = 0;
    for (int smp = 0; smp < 2000; ++smp) {
        const int r = (int)((uint64_t)smp * 2654435761ull % M), c = (int)((uint64_t)smp * 40503ull % N);
        double ref = 0, sabs = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) {
            const double p = (double)hA[(size_t)r * K + k] * hW[(size_t)c * K + k];
            ref += p; sabs += fabs(p);
            f = fmaf(hA[(size_t)r * K + k], hW[(size_t)c * K + k], f);
        }
        e_emul += fabs(hC[(size_t)r * N + c] - ref); e_f32 += fabs((double)f - ref); scale += sabs;
    }
    printf("%-26s terms=%d  %.3f ms  %6.1f TFLOP/s   err/sum|ab|: emulated %.3e  fp32 fmaf chain %.3e\n", tag, TERMS, ms,
           2.0 * M * N * K / ms / 1e9, e_emul / scale, e_f32 / scale);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50176, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 512;
    printf("M=%d N=%d K=%d\n", M, N, K);
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (auto& v : hA) v = rnd() * expf(3.f * rnd());
    for (auto& v : hW) v = rnd() * 0.1f;
    float *dA, *dW, *dC; uint16_t* dWp;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4)); CK(hipMalloc(&dWp, hW.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    const int64_t nW = (int64_t)N * K;
    split_planes<<<(nW + 255) / 256, 256>>>(dW, dWp, nW);
    run<6, 128, 128, 64, 32, 1>("128x128 bk32 single", dA, dWp, dC, M, N, K, hC, hA, hW);
    run<6, 256, 128, 64, 32, 2>("256x128 bk32 double", dA, dWp, dC, M, N, K, hC, hA, hW);
    run<6, 256, 128, 64, 32, 2, 1>("256x128 bk32 double sched", dA, dWp, dC, M, N, K, hC, hA, hW);
    run<6, 256, 256, 128, 16, 2>("256x256 bk16 double", dA, dWp, dC, M, N, K, hC, hA, hW);
    run<6, 256, 256, 128, 32, 1>("256x256 bk32 single", dA, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 256, 128, 64, 32, 2>("256x128 bk32 double", dA, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 256, 256, 128, 16, 2>("256x256 bk16 double", dA, dWp, dC, M, N, K, hC, hA, hW);
    return 0;
}
