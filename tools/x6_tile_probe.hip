// Probe for round 4: the production conv kernel's K loop (fp32 tiles global -> LDS by buffer LDS-DMA, un-padded [row][32 floats]
// with XOR-swizzled 16-byte slots, double buffer, one barrier per chunk) with three inner loops on the SAME tiles:
//   MODE 0  v_mfma_f32_32x32x2_f32 on the fp32 fragments (what conv_gemm_glds_kernel does today)
//   MODE 1  fp32 emulated on the bf16 matrix cores: every fragment is split IN REGISTERS into three bf16 planes (8+8+8 significant
//           bits, exact), six products hh, hm, mh, hl, lh, mm on v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   MODE 2  as 1, but the B operand (weights) arrives pre-split: three bf16 planes in HBM, fetched by LDS-DMA (6 B per element)
//   MODE 3  as 2 on v_mfma_f32_16x16x32_bf16
// C[M][N] = A[M][K] * W[N][K]^T, K % 32 == 0.  Prints TFLOP/s (2MNK) and the error against fp64 beside an fp32 fmaf chain's.
//   hipcc --offload-arch=gfx950 -O3 tools/x6_tile_probe.hip -o gpurun_out/x6tile && gpurun_out/x6tile [M N K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BK = 32;
__device__ unsigned long long* g_stamps = nullptr;   // per block: {cycles, 100 MHz ticks} around the K loop (diagnostic build only)
#ifndef STAMPS
#define STAMPS 0
#endif
constexpr unsigned DMA_OOB = 0xffffffffu;
typedef __amdgpu_buffer_rsrc_t dma_rsrc_t;
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffff00u, 0x00020000);
}
__device__ __forceinline__ void glds16b(dma_rsrc_t rs, unsigned voff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}

#ifndef SPLIT_TRUNC
#define SPLIT_TRUNC 0
#endif
// two floats -> three packed bf16 pairs (x0 in the low half)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
#if SPLIT_TRUNC
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
#else
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 hv = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(uint32_t, hv);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    bf2 mv = {(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(uint32_t, mv);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    bf2 lv = {(__bf16)s0, (__bf16)s1};
    l = __builtin_bit_cast(uint32_t, lv);
#endif
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& H, u32x4& M, u32x4& L) {
    uint32_t h, m, l;
    split2(a[0], a[1], h, m, l); H[0] = h; M[0] = m; L[0] = l;
    split2(a[2], a[3], h, m, l); H[1] = h; M[1] = m; L[1] = l;
    split2(b[0], b[1], h, m, l); H[2] = h; M[2] = m; L[2] = l;
    split2(b[2], b[3], h, m, l); H[3] = h; M[3] = m; L[3] = l;
}

__global__ void split_planes(const float* __restrict__ w, uint16_t* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, m, l;
    split2(w[i], w[i + 1], h, m, l);
    *(uint32_t*)(out + i) = h; *(uint32_t*)(out + n + i) = m; *(uint32_t*)(out + 2 * n + i) = l;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// MODE 3 (16x16x32 fragments: a lane holds k = 8 (lane / 16) .. + 7 of one of 16 rows): a ds_read_b128 lane group mixes two k groups
// (rows 0-3, 12-15 of k group g with rows 4-11 of k group g + 1), so the slot swizzles are chosen such that the rows {0-3, 12-15} map onto a
// slot set that is closed under XOR with the distance of the two k groups (2 fp32 slots / 1 plane slot): conflict-free by enumeration.
__device__ __forceinline__ int swzA16(int r) { const int q = (r >> 1) & 7; return (q & 4) | ((q & 1) << 1) | ((q >> 1) & 1); }
__device__ __forceinline__ int swzB16(int r) { const int q = (r >> 2) & 3; return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1); }

template <int MODE, int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_tile(const float* __restrict__ A, const float* __restrict__ W, const uint16_t* __restrict__ Wp,
                                                    float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AR = BM / 32;                       // A pieces per thread per chunk
    constexpr bool BPRE = MODE >= 2;
    constexpr int BR = BPRE ? 3 * BN / 64 : BN / 32;  // B pieces per thread per chunk (a bf16 plane row is 64 B: 16 rows per piece)
    constexpr int ABYTES = BM * BK * 4, BBYTES = BPRE ? 3 * BN * BK * 2 : BN * BK * 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* As0 = smem;
    uint8_t* Bs0 = smem + 2 * ABYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int lrow = tid >> 3, slot = tid & 7;
    unsigned adma[AR], bdma[BR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = lrow + 32 * j;
        const int ks = (slot ^ (MODE == 3 ? swzA16(row) : ((row >> 1) & 7))) * 4;
        adma[j] = (m0 + row) < M ? (unsigned)(((size_t)(m0 + row) * K + ks) * 4) : DMA_OOB;
    }
    if constexpr (!BPRE) {
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int row = lrow + 32 * j;
            const int ks = (slot ^ ((row >> 1) & 7)) * 4;
            bdma[j] = (unsigned)(((size_t)(n0 + row) * K + ks) * 4);
        }
    } else {
        // plane tile [BN][32 bf16]: thread -> row tid/4 (+64 per piece), 16-byte slot tid%4, swizzled by (row>>2)&3
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int pl = j / (BN / 64), jj = j % (BN / 64);
            const int row = (tid >> 2) + 64 * jj;
            const int sl = (tid & 3) ^ (MODE == 3 ? swzB16(row) : ((row >> 2) & 3));
            bdma[j] = (unsigned)(((size_t)pl * N * K + (size_t)(n0 + row) * K + sl * 8) * 2);
        }
    }
    int k0 = 0;
    auto fetch = [&](int buf) {
        uint8_t* la = As0 + buf * ABYTES + wave * 1024;
        uint8_t* lb = Bs0 + buf * BBYTES + wave * 1024;
        const dma_rsrc_t ra = dma_rsrc(A + k0);
#pragma unroll
        for (int j = 0; j < AR; ++j) glds16b(ra, adma[j], la + j * 4096);
        if constexpr (!BPRE) {
            const dma_rsrc_t rb = dma_rsrc(W + k0);
#pragma unroll
            for (int j = 0; j < BR; ++j) glds16b(rb, bdma[j], lb + j * 4096);
        } else {
            const dma_rsrc_t rb = dma_rsrc(Wp + k0);
#pragma unroll
            for (int j = 0; j < BR; ++j) glds16b(rb, bdma[j], lb + j * 4096);
        }
        k0 += BK;
    };

    constexpr int NACC = (MODE == 3) ? 4 : 1;         // 16x16 accumulators per 32x32 tile
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    (void)NACC;

    const int nchunks = K / BK;
    const int arow = wm * (BM / WM) + (lane & 31), brow = wn * (BN / WN) + (lane & 31);
    const int kh = lane >> 5;
    int aoff[TM], boff[TN], asw[TM], bsw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = arow + i * 32; aoff[i] = r * BK * 4; asw[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = brow + j * 32; boff[j] = BPRE ? r * 64 : r * BK * 4; bsw[j] = BPRE ? (r >> 2) & 3 : (r >> 1) & 7; }

    auto mma_chunk = [&](int buf) {
        const uint8_t* a = As0 + buf * ABYTES;
        const uint8_t* b = Bs0 + buf * BBYTES;
        if constexpr (MODE == 3) {
            // one k32 step per chunk on v_mfma_f32_16x16x32_bf16: the 32x32 tile (i, j) is four 16x16 accumulators, registers 4 (2 ii + jj) ..
            typedef __attribute__((ext_vector_type(4))) float f32x4a;
            const int l15 = lane & 15, kg = lane >> 4;
            u32x4 ah[TM][2], am[TM][2], al[TM][2], bh[TN][2], bm[TN][2], bl[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int r = wm * (BM / WM) + i * 32 + ii * 16 + l15, sw = swzA16(r);
                    const f32x4 x = *(const f32x4*)(a + r * (BK * 4) + (((2 * kg) ^ sw) << 4));
                    const f32x4 y = *(const f32x4*)(a + r * (BK * 4) + (((2 * kg + 1) ^ sw) << 4));
                    if constexpr (ABL & 1) { ah[i][ii] = __builtin_bit_cast(u32x4, x); am[i][ii] = __builtin_bit_cast(u32x4, y); al[i][ii] = ah[i][ii] ^ am[i][ii]; }
                    else split8(x, y, ah[i][ii], am[i][ii], al[i][ii]);
                }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int r = wn * (BN / WN) + j * 32 + jj * 16 + l15;
                    const unsigned o = (unsigned)(r * 64 + ((kg ^ swzB16(r)) << 4));
                    bh[j][jj] = *(const u32x4*)(b + o);
                    bm[j][jj] = *(const u32x4*)(b + BN * 64 + o);
                    bl[j][jj] = *(const u32x4*)(b + 2 * BN * 64 + o);
                }
#define MF16(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, X), __builtin_bit_cast(bf16x8, Y), Cc, 0, 0, 0)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            f32x4a c;
#pragma unroll
                            for (int r = 0; r < 4; ++r) c[r] = acc[i][j][(ii * 2 + jj) * 4 + r];
                            MF16(ah[i][ii], bl[j][jj], c); MF16(al[i][ii], bh[j][jj], c); MF16(am[i][ii], bm[j][jj], c);
                            MF16(ah[i][ii], bm[j][jj], c); MF16(am[i][ii], bh[j][jj], c); MF16(ah[i][ii], bh[j][jj], c);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][(ii * 2 + jj) * 4 + r] = c[r];
                        }
#undef MF16
        } else if constexpr (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f32x4 af[TM], bf[TN];
                const int q = ks * 2 + kh;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(a + aoff[i] + ((q ^ asw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4*)(b + boff[j] + ((q ^ bsw[j]) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {               // two k16 steps per chunk: this lane's 8 consecutive k = 16 s + 8 kh ..
                u32x4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
                const int q = s * 4 + kh * 2;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f32x4 x = *(const f32x4*)(a + aoff[i] + ((q ^ asw[i]) << 4));
                    const f32x4 y = *(const f32x4*)(a + aoff[i] + (((q + 1) ^ asw[i]) << 4));
                    if constexpr (ABL & 1) { ah[i] = __builtin_bit_cast(u32x4, x); am[i] = __builtin_bit_cast(u32x4, y); al[i] = ah[i] ^ am[i]; }
                    else split8(x, y, ah[i], am[i], al[i]);
                }
                if constexpr (!BPRE) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const f32x4 x = *(const f32x4*)(b + boff[j] + ((q ^ bsw[j]) << 4));
                        const f32x4 y = *(const f32x4*)(b + boff[j] + (((q + 1) ^ bsw[j]) << 4));
                        if constexpr (ABL & 1) { bh[j] = __builtin_bit_cast(u32x4, x); bm[j] = __builtin_bit_cast(u32x4, y); bl[j] = bh[j] ^ bm[j]; }
                        else split8(x, y, bh[j], bm[j], bl[j]);
                    }
                } else {
                    const int qb = s * 2 + kh;          // 16-byte slot of the plane row: 8 bf16
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[j] = *(const u32x4*)(b + boff[j] + ((qb ^ bsw[j]) << 4));
                        bm[j] = *(const u32x4*)(b + BN * 64 + boff[j] + ((qb ^ bsw[j]) << 4));
                        bl[j] = *(const u32x4*)(b + 2 * BN * 64 + boff[j] + ((qb ^ bsw[j]) << 4));
                    }
                }
#define MF(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, X), __builtin_bit_cast(bf16x8, Y), Cc, 0, 0, 0)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        MF(ah[i], bl[j], c); MF(al[i], bh[j], c); MF(am[i], bm[j], c);
                        MF(ah[i], bm[j], c); MF(am[i], bh[j], c); MF(ah[i], bh[j], c);
                        acc[i][j] = c;
                    }
#undef MF
            }
        }
    };
    unsigned long long st0 = 0, rt0 = 0;
    if (STAMPS) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    if (nchunks > 0) fetch(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (!(ABL & 2) && c + 1 < nchunks) fetch(buf ^ 1);
        mma_chunk((ABL & 2) ? 0 : buf);
        if (!(ABL & 4)) __syncthreads();
    }
    if (STAMPS && tid == 0) { g_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0; g_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0; }
    // epilogue through LDS: 16-byte row-contiguous stores (as store_tile_rows does)
    float* T = (float*)smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (MODE == 3) {      // register 4 (2 ii + jj) + q: row 16 ii + 4 (lane / 16) + q, column 16 jj + lane % 16
                const int ii = r >> 3, jj = (r >> 2) & 1, q = r & 3;
                const int row = wm * (BM / WM) + i * 32 + ii * 16 + 4 * (lane >> 4) + q;
#pragma unroll
                for (int j = 0; j < TN; ++j) T[row * BN + wn * (BN / WN) + j * 32 + jj * 16 + (lane & 15)] = acc[i][j][r];
            } else {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < TN; ++j) T[row * BN + wn * (BN / WN) + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        }
    __syncthreads();
    for (int e = tid; e < BM * BN / 4; e += 256) {
        const int row = e / (BN / 4), c4 = e % (BN / 4);
        if (m0 + row < M) *(f32x4*)(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = *(const f32x4*)(T + row * BN + c4 * 4);
    }
}


// MODE 4: the same arithmetic as MODE 1 as a three-stage software pipeline inside every wave.  Phase P(t), t = 2 * chunk + k16 step:
//   M(t)   the 6 * TM * TN MFMAs of step t on the bf16 planes made during P(t-1)
//   S(t+1) splits the raw fp32 fragments of step t + 1 (read at the head of this phase) into the other plane set
//   R(t+1) ds_read_b128 of step t + 1's fragments
// and one slot = one MFMA + its share of the side work, pinned with sched_barrier.  LDS: the two-buffer ring of the production kernel;
// the barrier sits at the head of the odd phases (chunk c + 1 landed, every wave has read chunk c), behind it the DMA of chunk c + 2.
template <int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_tile_pipe(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, F = TM + TN, NM = 6 * TM * TN, NU = 4 * F;
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int ABYTES = BM * BK * 4, BBYTES = BN * BK * 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* As0 = smem;
    uint8_t* Bs0 = smem + 2 * ABYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int lrow = tid >> 3, slot = tid & 7;
    unsigned adma[AR], bdma[BR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = lrow + 32 * j;
        adma[j] = (m0 + row) < M ? (unsigned)(((size_t)(m0 + row) * K + (slot ^ ((row >> 1) & 7)) * 4) * 4) : DMA_OOB;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int row = lrow + 32 * j;
        bdma[j] = (unsigned)(((size_t)(n0 + row) * K + (slot ^ ((row >> 1) & 7)) * 4) * 4);
    }
    const int nchunks = K / BK;
    auto fetchA = [&](int chunk, int buf) {
        uint8_t* la = As0 + buf * ABYTES + wave * 1024;
        const dma_rsrc_t ra = dma_rsrc(A + chunk * BK);
#pragma unroll
        for (int j = 0; j < AR; ++j) glds16b(ra, adma[j], la + j * 4096);
    };
    auto fetchB = [&](int chunk, int buf) {
        uint8_t* lb = Bs0 + buf * BBYTES + wave * 1024;
        const dma_rsrc_t rb = dma_rsrc(W + chunk * BK);
#pragma unroll
        for (int j = 0; j < BR; ++j) glds16b(rb, bdma[j], lb + j * 4096);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment f: A tile f (f < TM) or B tile f - TM; LDS byte address of its first 16-byte slot for step 0, buffer 0
    const int kh = lane >> 5;
    unsigned fbase[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const bool isA = f < TM;
        const int r = isA ? wm * (BM / WM) + (lane & 31) + f * 32 : wn * (BN / WN) + (lane & 31) + (f - TM) * 32;
        const int q = kh * 2;
        fbase[f] = (isA ? 0u : (unsigned)(2 * ABYTES)) + (unsigned)(r * BK * 4) + (unsigned)((q ^ ((r >> 1) & 7)) << 4);
    }
    f32x4 raw[F][2];
    u32x4 pl[2][F][3];
    auto rd = [&](int f, int half, int s, int buf) {      // step s: slots q + 4 s -> byte offset ^ 64; second half: ^ 16
        const unsigned o = (fbase[f] ^ (unsigned)(s * 64) ^ (unsigned)(half * 16)) + (unsigned)(buf * (f < TM ? ABYTES : BBYTES));
        raw[f][half] = *(const f32x4*)(smem + o);
    };
    auto unit = [&](int u, int set) {
        const int f = u >> 2, q = u & 3;
        uint32_t h, m, l;
        // pure ALU work floats freely in the IR (sched_barrier only binds the machine scheduler): an empty volatile asm on the inputs pins the unit to its slot
        { float x0 = raw[f][q >> 1][(q & 1) * 2], x1 = raw[f][q >> 1][(q & 1) * 2 + 1]; asm volatile("" : "+v"(x0), "+v"(x1)); raw[f][q >> 1][(q & 1) * 2] = x0; raw[f][q >> 1][(q & 1) * 2 + 1] = x1; }
        if constexpr (ABL & 1) { h = __float_as_uint(raw[f][q >> 1][(q & 1) * 2]); m = __float_as_uint(raw[f][q >> 1][(q & 1) * 2 + 1]); l = h ^ m; }
        else split2(raw[f][q >> 1][(q & 1) * 2], raw[f][q >> 1][(q & 1) * 2 + 1], h, m, l);
        asm volatile("" : "+v"(h), "+v"(m), "+v"(l));        // ... and one on the outputs keeps it from sinking towards its first use
        pl[set][f][0][q] = h; pl[set][f][1][q] = m; pl[set][f][2][q] = l;
    };
#define MFP(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, X), __builtin_bit_cast(bf16x8, Y), Cc, 0, 0, 0)
    auto mf = [&](int m, int set) {
        const int tp = m / 6, pr = m % 6, i = tp / TN, j = tp % TN;
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
        MFP(pl[set][i][PA[pr]], pl[set][TM + j][PB[pr]], acc[i][j]);
    };
    auto phase = [&](auto ph, int t) {
        constexpr int PH = decltype(ph)::value;
        constexpr int s = PH & 1, buf = PH >> 1, cur = PH & 1;
        constexpr int s1 = s ^ 1, b1 = s ? (buf ^ 1) : buf;        // step / buffer of t + 1
        if constexpr (s == 1) {
            if constexpr (!(ABL & 4)) __syncthreads();              // vmcnt(0) + lgkmcnt(0) + s_barrier
            if (!(ABL & 2) && (t + 3) / 2 < nchunks) fetchA((t + 3) / 2, buf);
        } else {
            if (!(ABL & 2) && t >= 2 && (t + 2) / 2 < nchunks) fetchB((t + 2) / 2, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mf(m, cur);
            if (m < 2 * F) rd(m >> 1, m & 1, s1, b1);
            // units of S(t+1): NU units over slots 4 .. NM-1
            constexpr int U0 = 4;
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (U0 + (u * (NM - U0)) / NU == m) unit(u, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;
    unsigned long long st0 = 0, rt0 = 0;
    if (STAMPS) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    if (nchunks > 0) { fetchA(0, 0); fetchB(0, 0); }
    if (nchunks > 1) { fetchA(1, 1); fetchB(1, 1); }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < F; ++f) { rd(f, 0, 0, 0); rd(f, 1, 0, 0); }
#pragma unroll
    for (int u = 0; u < NU; ++u) unit(u, 0);
    const int nph = 2 * nchunks;
    for (int t = 0; t < nph; t += 4) {
        phase(P0{}, t);
        phase(P1{}, t + 1);
        if (t + 2 < nph) {
            phase(P2{}, t + 2);
            phase(P3{}, t + 3);
        }
    }
    __syncthreads();
    if (STAMPS && tid == 0) { g_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0; g_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0; }
    float* T = (float*)smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < TN; ++j) T[row * BN + wn * (BN / WN) + j * 32 + (lane & 31)] = acc[i][j][r];
        }
    __syncthreads();
    for (int e = tid; e < BM * BN / 4; e += 256) {
        const int row = e / (BN / 4), c4 = e % (BN / 4);
        if (m0 + row < M) *(f32x4*)(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = *(const f32x4*)(T + row * BN + c4 * 4);
    }
}


// MODE 5: MODE 4's pipeline with the B operand as pre-split bf16 planes (LDS-DMA of 6 B per element, fragments are plain ds_read_b128 of
// the three planes: no VALU for B) and WM x WN waves of any count (NT = 64 * WM * WN threads).  With WN = 1 every A element is split by
// exactly one wave.
template <int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_tile_pipe2(const float* __restrict__ A, const uint16_t* __restrict__ Wp, float* __restrict__ C, int M, int N, int K) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NM = 6 * TM * TN, NU = 4 * TM;
    constexpr int APIECES = BM / 8, BPIECES = 3 * BN / 16;               // 1 KiB pieces per chunk
    constexpr int AR = (APIECES + NW - 1) / NW, BR = (BPIECES + NW - 1) / NW;
    constexpr int ABYTES = BM * BK * 4, BPLANE = BN * BK * 2, BBYTES = 3 * BPLANE;
    static_assert(APIECES % NW == 0 && BPIECES % NW == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* As0 = smem;
    uint8_t* Bs0 = smem + 2 * ABYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    unsigned adma[AR], bdma[BR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = (wave + NW * j) * 8 + (lane >> 3), sl = lane & 7;
        adma[j] = (m0 + row) < M ? (unsigned)(((size_t)(m0 + row) * K + (sl ^ ((row >> 1) & 7)) * 4) * 4) : DMA_OOB;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int pc = wave + NW * j, pl = pc / (BN / 16), q = pc % (BN / 16);
        const int row = q * 16 + (lane >> 2), sl = (lane & 3) ^ ((row >> 2) & 3);
        bdma[j] = (unsigned)(((size_t)pl * N * K + (size_t)(n0 + row) * K + sl * 8) * 2);
    }
    const int nchunks = K / BK;
    auto fetchA = [&](int chunk, int buf) {
        const dma_rsrc_t ra = dma_rsrc(A + chunk * BK);
#pragma unroll
        for (int j = 0; j < AR; ++j) glds16b(ra, adma[j], As0 + buf * ABYTES + (wave + NW * j) * 1024);
    };
    auto fetchB = [&](int chunk, int buf) {
        const dma_rsrc_t rb = dma_rsrc(Wp + chunk * BK);
#pragma unroll
        for (int j = 0; j < BR; ++j) glds16b(rb, bdma[j], Bs0 + buf * BBYTES + (wave + NW * j) * 1024);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kh = lane >> 5;
    unsigned abase[TM], bbase[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / WM) + (lane & 31) + i * 32;
        abase[i] = (unsigned)(r * BK * 4) + (unsigned)(((kh * 2) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / WN) + (lane & 31) + j * 32;
        bbase[j] = (unsigned)(2 * ABYTES) + (unsigned)(r * 64) + (unsigned)((kh ^ ((r >> 2) & 3)) << 4);
    }
    f32x4 raw[TM][2];
    u32x4 pa[2][TM][3], pb[2][TN][3];
    auto rdA = [&](int i, int half, int s, int buf) {
        raw[i][half] = *(const f32x4*)(smem + ((abase[i] ^ (unsigned)(s * 64) ^ (unsigned)(half * 16)) + (unsigned)(buf * ABYTES)));
    };
    auto rdB = [&](int j, int pl, int s, int buf, int set) {          // step s: plane slot kh + 2 s -> byte offset ^ 32
        pb[set][j][pl] = *(const u32x4*)(smem + ((bbase[j] ^ (unsigned)(s * 32)) + (unsigned)(buf * BBYTES + pl * BPLANE)));
    };
    auto unit = [&](int u, int set) {
        const int f = u >> 2, q = u & 3;
        uint32_t h, m, l;
        { float x0 = raw[f][q >> 1][(q & 1) * 2], x1 = raw[f][q >> 1][(q & 1) * 2 + 1]; asm volatile("" : "+v"(x0), "+v"(x1)); raw[f][q >> 1][(q & 1) * 2] = x0; raw[f][q >> 1][(q & 1) * 2 + 1] = x1; }
        if constexpr (ABL & 1) { h = __float_as_uint(raw[f][q >> 1][(q & 1) * 2]); m = __float_as_uint(raw[f][q >> 1][(q & 1) * 2 + 1]); l = h ^ m; }
        else split2(raw[f][q >> 1][(q & 1) * 2], raw[f][q >> 1][(q & 1) * 2 + 1], h, m, l);
        asm volatile("" : "+v"(h), "+v"(m), "+v"(l));
        pa[set][f][0][q] = h; pa[set][f][1][q] = m; pa[set][f][2][q] = l;
    };
    auto mf = [&](int m, int set) {
        const int tp = m / 6, pr = m % 6, i = tp / TN, j = tp % TN;
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
        MFP(pa[set][i][PA[pr]], pb[set][j][PB[pr]], acc[i][j]);
    };
    constexpr int NRD = 2 * TM + 3 * TN;                                 // LDS reads per step
    static_assert(NRD <= NM, "one read per slot");
    unsigned long long w_vm = 0, w_bar = 0;
    auto phase = [&](auto ph, int t) {
        constexpr int PH = decltype(ph)::value;
        constexpr int s = PH & 1, buf = PH >> 1, cur = PH & 1;
        constexpr int s1 = s ^ 1, b1 = s ? (buf ^ 1) : buf;
        if constexpr (s == 1) {
            if constexpr (ABL & 16) {          // where does a wave wait: for its own LDS-DMA (vmcnt) or for the other waves (barrier)?
                const unsigned long long q0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long q1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned long long q2 = __builtin_amdgcn_s_memtime();
                w_vm += q1 - q0; w_bar += q2 - q1;
            } else if constexpr (!(ABL & 4)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            if (!(ABL & 2) && (t + 3) / 2 < nchunks) { fetchA((t + 3) / 2, buf); if constexpr (ABL & 8) fetchB((t + 3) / 2, buf); }
        } else {
            if constexpr (!(ABL & 8)) if (!(ABL & 2) && t >= 2 && (t + 2) / 2 < nchunks) fetchB((t + 2) / 2, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mf(m, cur);
            if (m < 2 * TM) rdA(m >> 1, m & 1, s1, b1);
            else if (m < NRD) rdB((m - 2 * TM) / 3, (m - 2 * TM) % 3, s1, b1, cur ^ 1);
            constexpr int U0 = 2 * TM + 2;
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (U0 + (u * (NM - U0)) / NU == m) unit(u, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;
    unsigned long long st0 = 0, rt0 = 0;
    if (STAMPS) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    if (nchunks > 0) { fetchA(0, 0); fetchB(0, 0); }
    if (nchunks > 1) { fetchA(1, 1); fetchB(1, 1); }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { rdA(i, 0, 0, 0); rdA(i, 1, 0, 0); }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) rdB(j, pl, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) unit(u, 0);
    const int nph = 2 * nchunks;
    for (int t = 0; t < nph; t += 4) {
        phase(P0{}, t);
        phase(P1{}, t + 1);
        if (t + 2 < nph) {
            phase(P2{}, t + 2);
            phase(P3{}, t + 3);
        }
    }
    __syncthreads();
    if (STAMPS && tid == 0) { g_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0; g_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0; }
    if ((ABL & 16) && tid == 0) { g_stamps[2 * 65536 + 2 * blockIdx.x] = w_vm; g_stamps[2 * 65536 + 2 * blockIdx.x + 1] = w_bar; }
    float* T = (float*)smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < TN; ++j) T[row * BN + wn * (BN / WN) + j * 32 + (lane & 31)] = acc[i][j][r];
        }
    __syncthreads();
    for (int e = tid; e < BM * BN / 4; e += NT) {
        const int row = e / (BN / 4), c4 = e % (BN / 4);
        if (m0 + row < M) *(f32x4*)(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = *(const f32x4*)(T + row * BN + c4 * 4);
    }
}

template <int MODE, int BM, int BN, int WM, int WN, int ABL = 0>
static void run(const char* tag, const float* dA, const float* dW, const uint16_t* dWp, float* dC, int M, int N, int K, std::vector<float>& hC,
                const std::vector<float>& hA, const std::vector<float>& hW) {
    if (N % BN) { printf("%-28s skipped (N %% %d)\n", tag, BN); return; }
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    constexpr int BB = (MODE == 2 || MODE == 3 || MODE == 5) ? 3 * BN * BK * 2 : BN * BK * 4;
    int shm = 2 * (BM * BK * 4 + BB);
    if (shm < BM * BN * 4) shm = BM * BN * 4;
    if constexpr (MODE < 4) CK(hipFuncSetAttribute((const void*)gemm_tile<(MODE >= 4 ? 1 : MODE), BM, BN, WM, WN, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if constexpr (MODE == 4) CK(hipFuncSetAttribute((const void*)gemm_tile_pipe<BM, BN, WM, WN, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    if constexpr (MODE == 5) CK(hipFuncSetAttribute((const void*)gemm_tile_pipe2<BM, BN, WM, WN, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    auto launch = [&]() {
        if constexpr (MODE == 5) gemm_tile_pipe2<BM, BN, WM, WN, ABL><<<grid, 64 * WM * WN, shm>>>(dA, dWp, dC, M, N, K);
        else if constexpr (MODE == 4) gemm_tile_pipe<BM, BN, WM, WN, ABL><<<grid, 256, shm>>>(dA, dW, dC, M, N, K);
        else gemm_tile<(MODE >= 4 ? 1 : MODE), BM, BN, WM, WN, ABL><<<grid, 256, shm>>>(dA, dW, dWp, dC, M, N, K);
    };
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
    double clk = 0, cyc_chunk = 0;
    if (STAMPS) {
        std::vector<unsigned long long> hs(2 * (size_t)grid);
        unsigned long long* dptr; CK(hipMemcpyFromSymbol(&dptr, HIP_SYMBOL(g_stamps), sizeof(dptr)));
        CK(hipMemcpy(hs.data(), dptr, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cl, cy;
        for (int b = 0; b < grid; ++b) if (hs[2 * b + 1]) { cl.push_back((double)hs[2 * b] / hs[2 * b + 1] * 0.1); cy.push_back((double)hs[2 * b] / (K / BK)); }
        std::sort(cl.begin(), cl.end()); std::sort(cy.begin(), cy.end());
        if (!cl.empty()) { clk = cl[cl.size() / 2]; cyc_chunk = cy[cy.size() / 2]; }
    }
    double e_emul = 0, e_f32 = 0, scale = 0, e_max = 0;
    for (int smp = 0; smp < 3000; ++smp) {
        const int r = (int)((uint64_t)smp * 2654435761ull % M), c = (int)((uint64_t)smp * 40503ull % N);
        double ref = 0, sabs = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) {
            const double p = (double)hA[(size_t)r * K + k] * hW[(size_t)c * K + k];
            ref += p; sabs += fabs(p);
            f = fmaf(hA[(size_t)r * K + k], hW[(size_t)c * K + k], f);
        }
        const double e = fabs(hC[(size_t)r * N + c] - ref);
        e_emul += e; e_f32 += fabs((double)f - ref); scale += sabs;
        if (e / sabs > e_max) e_max = e / sabs;
    }
    printf("%-28s mode %d  %8.4f ms  %6.1f TFLOP/s   err/sum|ab|: kernel %.3e (max %.2e)  fp32 fmaf chain %.3e", tag, MODE, ms,
           2.0 * M * N * K / ms / 1e9, e_emul / scale, e_max, e_f32 / scale);
    if (STAMPS) printf("   clock %.2f GHz, %.0f cycles / chunk / block", clk, cyc_chunk);
    if (STAMPS && (ABL & 16)) {
        std::vector<unsigned long long> hw(2 * (size_t)grid);
        unsigned long long* dptr; CK(hipMemcpyFromSymbol(&dptr, HIP_SYMBOL(g_stamps), sizeof(dptr)));
        CK(hipMemcpy(hw.data(), dptr + 2 * 65536, hw.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> a, b;
        for (int q = 0; q < grid; ++q) { a.push_back((double)hw[2 * q] / (K / BK)); b.push_back((double)hw[2 * q + 1] / (K / BK)); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("   wave 0 per chunk: %.0f cycles waiting for its LDS-DMA, %.0f at the barrier", a[a.size() / 2], b[b.size() / 2]);
    }
    printf("\n");
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 100352, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 1152;
    printf("M=%d N=%d K=%d  split=%s\n", M, N, K, SPLIT_TRUNC ? "truncate" : "round-to-nearest");
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (auto& v : hA) v = rnd() * expf(3.f * rnd());
    for (auto& v : hW) v = rnd() * 0.1f;
    float *dA, *dW, *dC; uint16_t* dWp;
    CK(hipMalloc(&dA, hA.size() * 4 + 4096)); CK(hipMalloc(&dW, hW.size() * 4 + 4096)); CK(hipMalloc(&dC, hC.size() * 4)); CK(hipMalloc(&dWp, hW.size() * 6 + 4096));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    { unsigned long long* ds; CK(hipMalloc(&ds, 16 * 65536 * 8)); CK(hipMemset(ds, 0, 16 * 65536 * 8));   /* [0, 2 x 65536): loop stamps; [2 x 65536, ..): wait stamps */ CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds))); }
    const int64_t nW = (int64_t)N * K;
    split_planes<<<(nW / 2 + 255) / 256, 256>>>(dW, dWp, nW);
    CK(hipDeviceSynchronize());
    run<0, 128, 128, 2, 2>("128x128 fp32 mfma", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 128, 128, 2, 2>("128x128 x6 split A,B in regs", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<2, 128, 128, 2, 2>("128x128 x6 B planes", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<3, 128, 128, 2, 2>("128x128 x6 B planes 16x16x32", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<2, 128, 128, 2, 2>("128x128 x6 B planes (again)", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<3, 128, 128, 2, 2>("128x128 x6 B planes 16x16x32", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<2, 128, 128, 2, 2, 1>("  32x32x16 abl: no split", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<3, 128, 128, 2, 2, 1>("  16x16x32 abl: no split", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<2, 128, 128, 2, 2, 3>("  32x32x16 abl: no split/DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<3, 128, 128, 2, 2, 3>("  16x16x32 abl: no split/DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<4, 128, 128, 2, 2>("128x128 x6 pipelined", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<4, 128, 128, 2, 2, 1>("  pipe abl: no split", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<4, 128, 128, 2, 2, 2>("  pipe abl: no DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 128, 128, 4, 1>("128x128 pipe, B planes, 4x1", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 128, 128, 2, 2>("128x128 pipe, B planes, 2x2", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 8, 1>("256x128 pipe, B planes, 8x1", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 8, 1, 1>("  abl: no split", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 8, 1, 2>("  abl: no DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 4, 2>("256x128 pipe, B planes, 4x2", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 128, 64, 4, 1>("128x64 pipe, B planes, 4x1", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 128, 64, 4, 1, 8>("  early B fetch", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 128, 64, 4, 1, 16>("  wait stamps", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 128, 64, 4, 1, 24>("  early B + wait stamps", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 8, 1, 8>("256x128 early B fetch", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 8, 1, 24>("  early B + wait stamps", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<5, 256, 128, 8, 1, 16>("256x128 wait stamps", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 128, 128, 2, 2, 1>("  abl: no split VALU", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 128, 128, 2, 2, 2>("  abl: no DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 128, 128, 2, 2, 3>("  abl: no split, no DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<1, 128, 128, 2, 2, 7>("  abl: no split/DMA/barrier", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    run<0, 128, 128, 2, 2, 2>("  fp32 abl: no DMA", dA, dW, dWp, dC, M, N, K, hC, hA, hW);
    return 0;
}
