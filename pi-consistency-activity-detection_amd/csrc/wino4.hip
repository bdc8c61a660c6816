// Winograd F(4x4, 3x3) over (h, w) for the largest stride-1, pad-1 3x3x3 convolutions of the step and their input gradients (the decoder's
// skip convs conv112 / conv56, capsules_ucf101.py:382-384,497,501, both ways; the input gradient of Conv3d_2c at 56 x 56, pytorch_i3d.py:236-238
// -- its FORWARD and the 28 x 28 Inception branches stay in F(2x2, 3x3): they are the trunk's forward, in front of EM routing):
// 4x fewer multiply-accumulates than the direct form, 1.78x fewer than F(2x2, 3x3) (wino.hip), for about 3x wino.hip's rounding error
// (tests/test_wino4_gpu.py holds the same 2e-5 bar; tests/test_wino4_cpu.py restates the algebra below in fp64).  Same structure as wino.hip --
// ONE fused kernel, the transform-domain tensors never exist in HBM -- with the roles laid out for 36 transform positions instead of 16:
//
//   Y = A^T [ sum_ci sum_kt (G g_kt G^T) .* (B^T d_kt B) ] A       per 4x4 output tile, d = 6x6 input patch at (4i-1, 4j-1)
// Interpolation points (0, +a, -a, +b, -b, inf) with a = 1/sqrt(2), b = sqrt(2) instead of the textbook (0, +-1, +-2, inf): the same operation
// count (the +- pairs share their even and odd parts), half the rms and a quarter of the maximum rounding error on post-ReLU inputs --
// 1.9x the rms error of a plain fp32 accumulation chain instead of 4x (a scan over symmetric pairs: flat optimum around a = 0.65 - 0.71,
// b = 1.41 - 1.5; Barabasz et al., "Error analysis and improving the accuracy of Winograd convolution for deep neural networks"):
//   B^T = [a2b2 0 -(a2+b2) 0 1 0; 0 -+ab2 -b2 +-a 1 0 (p = +-a); 0 -+a2b -a2 +-b 1 0 (p = +-b); 0 a2b2 0 -(a2+b2) 0 1]
//   G   = [1/(a2b2) 0 0; (1 +-a a2) / (2a2(a2-b2)); (1 +-b b2) / (2b2(b2-a2)); 0 0 1]          A^T columns = (1 p p2 p3), last (0 0 0 1)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr unsigned DMA_OOB = 0xffffffffu;
typedef __amdgpu_buffer_rsrc_t dma_rsrc_t;
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffff00u, 0x00020000);
}
template <bool ON = true>
__device__ __forceinline__ void glds16b(dma_rsrc_t rs, unsigned voff, float* l) {
    if (ON) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

struct Wino4K {
    const float* in; const float* U; const float* bias; float* out; float* bnpart;
    int N, T, H, W, Ci, ldi, Co, ldo;
    int TH, TW, BTH, BTW, nbh, nbw, nct, nc4;
    int KT, act, flags;
    int Ti, ta, tc, tden;
    int btw_magic;
    int rpitch, rq;                         // raw-patch image: positions per patch row / positions per column class (c & 3) inside a row
};

constexpr float PA = 0.70710678118654752f, PB = 1.41421356237309505f;      // the interpolation points +-a, +-b
constexpr float PA2 = PA * PA, PB2 = PB * PB, PA3 = PA2 * PA, PB3 = PB2 * PB, P0 = PA2 * PB2, PS = PA2 + PB2;
constexpr int XT = 32;            // tiles per block (rows of the transform-domain GEMMs)
constexpr int XC = 64;            // output channels per block
constexpr int XK = 4;             // input channels per K chunk
// Operand images (V in LDS, U in global memory).  Transform position (xi, nu) has index P = 18 (nu / 3) + 3 xi + nu % 3: wave (nh, wn) owns the 18
// positions of its nu half nh for all 32 tiles x its 32 channels.  Two consecutive positions share a 16-byte slot, so one 16-byte read per lane is
// the fragment of four MFMAs:   V[P / 2][k half][tile 32][P % 2][2]      U[P / 2][k half][co 64][P % 2][2]      channel k of the chunk = 2 (k half) + e
constexpr int VPLANE = 18 * 2 * XT * 4;
constexpr int UPLANE = 18 * 2 * XC * 4;
constexpr int RPIECES = 12;       // 1 KiB LDS-DMA pieces of one raw patch image (64 positions x 4 channels each): up to 768 positions
constexpr int RPLANE = RPIECES * 256;

// ---- weight transform: U[kt][ct][c4][P / 2][k half][co 64][P % 2][2] = (G g G^T)[xi][nu] of g = w[o][kt][.][.][i], read through strides (see wino.hip)
__global__ void wino4_weights_kernel(const float* __restrict__ w, long long sO, long long sT, long long sI, int O, int I, int KT, int flip,
                                     float* __restrict__ U, int nct, int nc4) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)KT * nct * 64 * I;
    if (e >= total) return;
    const int i = (int)(e % I);
    long long r = e / I;
    const int o = (int)(r % (nct * 64));
    const int kt = (int)(r / (nct * 64));
    // in double, rounded once: the transform's constants are irrational, and U's rounding is shared by every tile of the layer
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ks = flip ? KT - 1 - kt : kt, as = flip ? 2 - a : a, bs = flip ? 2 - b : b;
            g[a][b] = o < O ? (double)w[(long long)o * sO + (long long)((ks * 3 + as) * 3 + bs) * sT + (long long)i * sI] : 0.0;
        }
    constexpr double A_ = 0.70710678118654752440, B_ = 1.41421356237309504880, A2 = 0.5, B2 = 2.0;
    constexpr double g0 = 1.0 / (A2 * B2), na = 1.0 / (2.0 * A2 * (A2 - B2)), nb = 1.0 / (2.0 * B2 * (B2 - A2));
    static_assert(PA2 > 0.4999f && PA2 < 0.5001f && PB2 > 1.9999f && PB2 < 2.0001f, "wino4_weights_kernel holds a = 1/sqrt(2), b = sqrt(2) in double");
    double t[6][3];          // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g0 * g[0][b];
        t[1][b] = na * (g[0][b] + A_ * g[1][b] + A2 * g[2][b]);
        t[2][b] = na * (g[0][b] - A_ * g[1][b] + A2 * g[2][b]);
        t[3][b] = nb * (g[0][b] + B_ * g[1][b] + B2 * g[2][b]);
        t[4][b] = nb * (g[0][b] - B_ * g[1][b] + B2 * g[2][b]);
        t[5][b] = g[2][b];
    }
    const int k = i & 3;
    float* dst = U + (((long long)kt * nct + o / 64) * nc4 + i / 4) * UPLANE + ((long long)(k >> 1) * 64 + (o & 63)) * 4 + (k & 1);
#pragma unroll
    for (int x = 0; x < 6; ++x) {
        double u[6];
        u[0] = g0 * t[x][0];
        u[1] = na * (t[x][0] + A_ * t[x][1] + A2 * t[x][2]);
        u[2] = na * (t[x][0] - A_ * t[x][1] + A2 * t[x][2]);
        u[3] = nb * (t[x][0] + B_ * t[x][1] + B2 * t[x][2]);
        u[4] = nb * (t[x][0] - B_ * t[x][1] + B2 * t[x][2]);
        u[5] = t[x][2];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) {
            const int P = (nu / 3) * 18 + x * 3 + nu % 3;
            dst[(P >> 1) * (2 * 64 * 4) + (P & 1) * 2] = (float)u[nu];
        }
    }
}

// ---- the fused convolution.  Block = 32 tiles (a BTH x BTW rectangle of 4x4-output tiles of one (n, t) plane) x 64 output channels; four
// waves, one per SIMD, as 2 (nu halves) x 2 (channel halves): a wave holds the 18 accumulators of its half of the transform positions for all
// 32 tiles x 32 channels (288 registers) and the two nu halves meet in the epilogue.  K chunk = 4 input channels of one temporal tap.
//   R  the raw (4 BTH + 2) x (4 BTW + 2) input patch of the block, [position][4 channels]: 16 B per position, zeros for padding, fetched once
//      per block by LDS-DMA, all twelve 1 KiB pieces from wave 3, which does no transform; position (r, c) sits at
//      r * rpitch + (c & 3) * rq + c / 4, so the 32 tiles of a ds_read_b128 read consecutive slots (rpitch chosen on the host so that tiles of
//      different tile rows do not meet either),
//   V  the transformed patch B^T d B, written by waves 0-2: thread = (tile, B^T row xi): 24 ds_read_b128 of R, the row stage as four multiply-adds
//      per column with per-lane coefficients, the column stage in its factored form, 6 ds_write2st64_b64,
//   U  the transformed weights never pass through LDS: every wave reads its own nine fragments of the next chunk (16 bytes per lane, the layout
//      wino4_weights_kernel writes is the MFMA's own) straight from global memory into registers, in the chunk's first nine MFMA gaps.  (The
//      first version sent U through LDS with 36 LDS-DMA pieces per chunk from wave 3: 3900 - 4030 cycles per chunk against 3120 - 3230 now,
//      profiles/r05_wino4_probe.txt.)
// R runs two chunks ahead, U one; one barrier per chunk; the loop is unrolled by two so that every LDS buffer and register set is a compile-time
// choice.  The two roles are two copies of the K loop (a branch inside the loop would split it into blocks and degrade every s_waitcnt); the
// barrier counts arrivals, not program counters.
// VAR (PICONS_DIAG builds; results are WRONG with bits 1-8): 1 = no patch DMA, 2 = no weight loads, 4 = no transform stores, 8 = no output stores,
// 32 = s_memtime stamps of prologue / K loop / epilogue per block into bnpart (4 x u64 per block), 64 = the product kernel with stamps around
// the epilogue's three phases as well (8 x u64 per block: prologue, loop, row/column stage + send, finalise, stores, chunks).
template <int VAR>
__global__ __launch_bounds__(256, 1) void wino4_conv_kernel(const Wino4K p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Rs = smem;                        // [2][RPLANE]
    float* Vs = smem + 2 * RPLANE;           // [2][VPLANE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wave >> 1, wn = wave & 1;
    int sb = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = sb % p.nct; sb /= p.nct;
    const int sblock = sb;             // spatial block id (n, t, bh, bw): the BatchNorm partial row pair
    const int bw = sb % p.nbw; sb /= p.nbw;
    const int bh = sb % p.nbh; sb /= p.nbh;
    const int t = sb % p.T, n = sb / p.T;
    const int PW = 4 * p.BTW + 2, PH = 4 * p.BTH + 2;
    const int h0 = 4 * bh * p.BTH - 1, w0 = 4 * bw * p.BTW - 1;          // image position of patch position (0, 0)

    // raw-patch DMA role (wave 3): all RPIECES pieces of the image; lane -> slot q = 64 piece + lane
    unsigned roff[RPIECES];
#pragma unroll
    for (int i = 0; i < RPIECES; ++i) {
        const int q = i * 64 + lane;
        const int pr = q / p.rpitch, rem = q - pr * p.rpitch;
        const int cm = rem / p.rq, idx = rem - cm * p.rq;
        const int pc = 4 * idx + cm;
        const int h = h0 + pr, w = w0 + pc;
        const bool ok = pr < PH && cm < 4 && pc < PW && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        roff[i] = ok ? (unsigned)(((h * p.W + w) * p.ldi) * 4) : DMA_OOB;
    }
    // weight fragments: straight from global memory (L2) into registers -- lane (k half, channel) of pair pp reads 16 bytes at
    // ((pp' * 2 + k half) * 64 + channel) * 16, pp' = 9 nh + pp: the layout the MFMA wants, so U never passes through LDS
    const unsigned uoff = (unsigned)(((nh * 18 + (lane >> 5)) * XC + wn * 32 + (lane & 31)) * 16);

    // transform role (waves 0-2): thread = (tile lane % 32, B^T row xi); waves 0 / 1 / 2 hold xi = (1, 2) / (3, 4) / (0, 5)
    const int ttile = lane & 31;
    const int xi = wave == 2 ? (lane >> 5) * 5 : 1 + 2 * (wave & 1) + (lane >> 5);
    const int tli = ttile / p.BTW, tlj = ttile - tli * p.BTW;
    const bool tval = ttile < p.BTH * p.BTW;
    // rows of the patch this B^T row reads and its coefficients: xi = 0: (0, 2, 4); xi = 5: (1, 3, 5); else (1, 2, 3, 4): p = +a, -a, +b, -b
    int prow[4];
    float ca[4];
    {
        const bool edge = xi == 0 || xi == 5;
        const int r0 = xi == 0 ? 0 : 1, rs = edge ? 2 : 1;
        prow[0] = r0; prow[1] = r0 + rs; prow[2] = r0 + 2 * rs; prow[3] = edge ? r0 + 2 * rs : r0 + 3;
        ca[0] = edge ? P0 : (xi == 1 ? -PA * PB2 : (xi == 2 ? PA * PB2 : (xi == 3 ? -PA2 * PB : PA2 * PB)));
        ca[1] = edge ? -PS : (xi <= 2 ? -PB2 : -PA2);
        ca[2] = edge ? 1.f : (xi == 1 ? PA : (xi == 2 ? -PA : (xi == 3 ? PB : -PB)));
        ca[3] = edge ? 0.f : 1.f;
    }
    int ro[4];                                       // float offset of (patch row prow[i], column 0) of this thread's tile in R
#pragma unroll
    for (int i = 0; i < 4; ++i) ro[i] = tval ? ((4 * tli + prow[i]) * p.rpitch + tlj) * 4 : 0;
    int colo[6];                                     // wave-uniform float offset of patch column c
#pragma unroll
    for (int c = 0; c < 6; ++c) colo[c] = ((c & 3) * p.rq + (c >> 2)) * 4;
    int vo[6];                                       // float offset of this thread's (xi, nu) slot in a V plane, k half 0
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {
        const int P = (nu / 3) * 18 + xi * 3 + nu % 3;
        vo[nu] = ((P >> 1) * 2 * XT + ttile) * 4 + (P & 1) * 2;
    }

    // temporal taps whose source frame exists (see wino.hip)
    int nkt = 0, ktl0 = 0, ktl1 = 0, ktl2 = 0, ttl0 = 0, ttl1 = 0, ttl2 = 0;
    for (int kt = 0; kt < p.KT; ++kt) {
        const int num = t * p.ta + kt + p.tc;
        if (num < 0 || num % p.tden) continue;
        const int tt = num / p.tden;
        if (tt >= p.Ti) continue;
        if (nkt == 0) { ktl0 = kt; ttl0 = tt; } else if (nkt == 1) { ktl1 = kt; ttl1 = tt; } else { ktl2 = kt; ttl2 = tt; }
        ++nkt;
    }
    const int nchunks = nkt * p.nc4;
    const size_t plane_in = (size_t)p.H * p.W * p.ldi;
    const float* rtap0 = p.in + ((size_t)n * p.Ti + ttl0) * plane_in;
    const float* rtap1 = p.in + ((size_t)n * p.Ti + ttl1) * plane_in;
    const float* rtap2 = p.in + ((size_t)n * p.Ti + ttl2) * plane_in;
    const float* utap0 = p.U + (((size_t)ktl0 * p.nct + ct) * p.nc4) * UPLANE;
    const float* utap1 = p.U + (((size_t)ktl1 * p.nct + ct) * p.nc4) * UPLANE;
    const float* utap2 = p.U + (((size_t)ktl2 * p.nct + ct) * p.nc4) * UPLANE;
    auto advance = [&](int& q, int& c4) {
        const bool wrap = c4 + 1 == p.nc4, last = wrap && q + 1 >= nkt;
        c4 = last ? c4 : (wrap ? 0 : c4 + 1);
        q = (wrap && !last) ? q + 1 : q;
    };
    auto r_base = [&](int q, int c4) -> const float* { return (q == 0 ? rtap0 : (q == 1 ? rtap1 : rtap2)) + c4 * XK; };
    auto u_base = [&](int q, int c4) -> const float* { return (q == 0 ? utap0 : (q == 1 ? utap1 : utap2)) + (size_t)c4 * UPLANE; };
    int uq = 0, uc4 = 0, rq_ = 0, rc4 = 0;

    const float* rnext = nullptr;
    const float* ug = nullptr;
    const float* rsrc = nullptr;
    float* vb = nullptr;
    f32x4 dA[4], dB[4], Tc[6], tq, eA, eB, eC, eD, w0v, w1v;
    f32x4 bq0[9], bq1[9];                                       // weight fragments of the chunk in flight / of the next one
    auto issue_r = [&](const float* base, int rb) {           // raw patch -> R[rb] (prologue; wave 3)
        const dma_rsrc_t rs = dma_rsrc(base);
#pragma unroll
        for (int i = 0; i < RPIECES; ++i) glds16b<!(VAR & 1)>(rs, roff[i], Rs + rb * RPLANE + i * 256);
    };
    auto load_b = [&](const float* base, int pp) -> f32x4 {
        if (VAR & 2) return (f32x4){0.f, 0.f, 0.f, 0.f};
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dma_rsrc(base), uoff, pp * (2 * XC * 16), 0));
    };
    auto readcol = [&](f32x4* d, int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = *(const f32x4*)(rsrc + ro[i] + colo[c]);
    };
    auto bc = [](float s) { return (f32x4){s, s, s, s}; };
    auto fma4 = [](f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); };
    auto vstore = [&](int nu, f32x4 v) {
        if (!(VAR & 4)) {
            *(f32x2*)(vb + vo[nu]) = (f32x2){v[0], v[1]};
            *(f32x2*)(vb + vo[nu] + XT * 4) = (f32x2){v[2], v[3]};
        }
    };
    // Side work of one K chunk, one piece per MFMA gap (step s = 4 * group + slot, 9 groups of 4 MFMAs = one position pair each).  Every wave
    // fetches its nine weight fragments of chunk c + 1 in the first nine gaps (a chunk for them to arrive).
    // ROLE 0 (transform): patch columns are read in the order (1, 2, 3, 4, 0, 5), one per group, into two alternating register sets; the row
    // stage of a column runs one group later (its reads were issued in front of that group's fragments, so the group's own lgkmcnt wait covers
    // them), the column stage as soon as its inputs exist.  ROLE 1 (wave 3): the raw-patch pieces of chunk c + 2, one per gap behind the fragments.
    auto side = [&](auto ROLE, auto S, auto PHASE) {
        constexpr int role = decltype(ROLE)::value, s_ = decltype(S)::value, buf = decltype(PHASE)::value;
        constexpr int g = s_ >> 2, e = s_ & 3;
        if constexpr (s_ < 9) (buf ? bq0 : bq1)[s_] = load_b(ug, s_);
        if constexpr (role == 1) {
            if constexpr (s_ >= 9 && s_ < 9 + RPIECES) glds16b<!(VAR & 1)>(dma_rsrc(rnext), roff[s_ - 9], Rs + buf * RPLANE + (s_ - 9) * 256);
        } else {
            constexpr int cols[6] = {1, 2, 3, 4, 0, 5};
            if constexpr (e == 0 && g < 6) readcol((g & 1) ? dB : dA, cols[g]);
            if constexpr (g >= 1 && g <= 6 && (e == 1 || e == 2)) {
                f32x4* d = ((g - 1) & 1) ? dB : dA;
                if constexpr (e == 1) tq = fma4(bc(ca[1]), d[1], bc(ca[0]) * d[0]);
                else Tc[cols[g - 1]] = fma4(bc(ca[3]), d[3], fma4(bc(ca[2]), d[2], tq));
            }
            if constexpr (g == 3 && e == 3) { eB = fma4(bc(-PB2), Tc[1], Tc[3]); eD = fma4(bc(-PA2), Tc[1], Tc[3]); }
            if constexpr (g == 4 && e == 3) { eA = fma4(bc(-PB2), Tc[2], Tc[4]); eC = fma4(bc(-PA2), Tc[2], Tc[4]); }
            if constexpr (g == 5 && e == 3) { w0v = fma4(bc(PA), eB, eA); w1v = fma4(bc(-PA), eB, eA); }
            if constexpr (g == 6 && e == 0) { vstore(1, w0v); vstore(2, w1v); }
            if constexpr (g == 6 && e == 3) { w0v = fma4(bc(PB), eD, eC); w1v = fma4(bc(-PB), eD, eC); }
            if constexpr (g == 7 && e == 0) { vstore(3, w0v); vstore(4, w1v); }
            if constexpr (g == 7 && e == 1) w0v = fma4(bc(P0), Tc[0], fma4(bc(-PS), Tc[2], Tc[4]));
            if constexpr (g == 7 && e == 2) w1v = fma4(bc(P0), Tc[1], fma4(bc(-PS), Tc[3], Tc[5]));
            if constexpr (g == 8 && e == 0) { vstore(0, w0v); vstore(5, w1v); }
        }
    };
    // 18 accumulators = 288 registers: 16 of them fill the 256 AGPRs, two live in VGPRs.  The MFMAs are inline asm with the register class
    // spelled out: left to itself hipcc gives every MFMA an AGPR destination and moves three accumulators between the files around their
    // MFMAs in every chunk (96 moves and three full MFMA-latency stalls per chunk).
    f32x16 acc[18];
#pragma unroll
    for (int i = 0; i < 18; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    if (VAR & (32 | 64)) stamp[0] = __builtin_amdgcn_s_memtime();

    const int kh = lane >> 5;
    const int aoff = ((nh * 18 + kh) * XT + (lane & 31)) * 4;
    // prologue: R(0), R(1) in flight (wave 3), every wave's weight fragments of chunk 0; V(0) from R(0)
    if (nchunks > 0) {
        if (wave == 3) {
            issue_r(r_base(rq_, rc4), 0);
            advance(rq_, rc4);
            issue_r(r_base(rq_, rc4), 1);
            advance(rq_, rc4);                       // -> chunk 2
        }
        ug = u_base(uq, uc4);
#pragma unroll
        for (int j = 0; j < 9; ++j) bq0[j] = load_b(ug, j);
        advance(uq, uc4);                            // -> chunk 1
    }
    PC_SYNC_DMA();
    if (nchunks > 0 && wave < 3) {
        rsrc = Rs;
        vb = Vs;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            readcol(dA, c);
            Tc[c] = fma4(bc(ca[3]), dA[3], fma4(bc(ca[2]), dA[2], fma4(bc(ca[1]), dA[1], bc(ca[0]) * dA[0])));
        }
        eB = fma4(bc(-PB2), Tc[1], Tc[3]); eD = fma4(bc(-PA2), Tc[1], Tc[3]);
        eA = fma4(bc(-PB2), Tc[2], Tc[4]); eC = fma4(bc(-PA2), Tc[2], Tc[4]);
        vstore(0, fma4(bc(P0), Tc[0], fma4(bc(-PS), Tc[2], Tc[4])));
        vstore(1, fma4(bc(PA), eB, eA));
        vstore(2, fma4(bc(-PA), eB, eA));
        vstore(3, fma4(bc(PB), eD, eC));
        vstore(4, fma4(bc(-PB), eD, eC));
        vstore(5, fma4(bc(P0), Tc[1], fma4(bc(-PS), Tc[3], Tc[5])));
    }
    __syncthreads();
    if (VAR & (32 | 64)) stamp[1] = __builtin_amdgcn_s_memtime();
    // One K chunk; PHASE = chunk parity (the loop is unrolled by two, so every LDS buffer and fragment set is a compile-time choice; nchunks is
    // even: Ci % 8 == 0).  During chunk c: weight fragments of c+1 -> the other register set, R(c+2) -> R[buf] (held chunk c, transformed one
    // iteration ago), V(c+1) from R[buf^1].
    auto chunk = [&](auto ROLE, auto PHASE) {
        constexpr int role = decltype(ROLE)::value, buf = decltype(PHASE)::value;
        ug = u_base(uq, uc4);
        advance(uq, uc4);
        if constexpr (role == 1) {
            rnext = r_base(rq_, rc4);
            advance(rq_, rc4);
        } else {
            rsrc = Rs + (buf ^ 1) * RPLANE;
            vb = Vs + (buf ^ 1) * VPLANE;
        }
        const float* va = Vs + buf * VPLANE + aoff;
        f32x4* bq = buf ? bq1 : bq0;
        f32x4 a0 = *(const f32x4*)va, a1;
#define W4_STEP(PP, E, A)                                                                                                      \
        if constexpr (2 * (PP) + ((E) >> 1) < 16)                                                                              \
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[2 * (PP) + ((E) >> 1)]) : "v"(A[E]), "v"(bq[PP][E])); \
        else                                                                                                                   \
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[2 * (PP) + ((E) >> 1)]) : "v"(A[E]), "v"(bq[PP][E])); \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        side(ROLE, std::integral_constant<int, 4 * (PP) + (E)>{}, PHASE);                                                      \
        __builtin_amdgcn_sched_barrier(0);
#define W4_GROUP(PP, A, AN)                                                                                                    \
        {                                                                                                                      \
            W4_STEP(PP, 0, A)                                                                                                  \
            if ((PP) + 1 < 9) AN = *(const f32x4*)(va + ((PP) + 1) * (2 * XT * 4));                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
            W4_STEP(PP, 1, A) W4_STEP(PP, 2, A) W4_STEP(PP, 3, A)                                                              \
        }
        W4_GROUP(0, a0, a1)   W4_GROUP(1, a1, a0)   W4_GROUP(2, a0, a1)
        W4_GROUP(3, a1, a0)   W4_GROUP(4, a0, a1)   W4_GROUP(5, a1, a0)
        W4_GROUP(6, a0, a1)   W4_GROUP(7, a1, a0)   W4_GROUP(8, a0, a1)
#undef W4_STEP
#undef W4_GROUP
        PC_SYNC_DMA();
    };
    auto k_loop = [&](auto ROLE) {
        for (int c = 0; c < nchunks; c += 2) {
            chunk(ROLE, std::integral_constant<int, 0>{});
            chunk(ROLE, std::integral_constant<int, 1>{});
        }
    };
    if (wave == 3) k_loop(std::integral_constant<int, 1>{});
    else k_loop(std::integral_constant<int, 0>{});
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs' results (inline asm: the compiler does not count their latency)
    __builtin_amdgcn_sched_barrier(0);                           // ... and nothing that reads an accumulator may be scheduled above the nops (a memory
                                                                 // clobber does not order register reads)

    if (VAR & (32 | 64)) stamp[2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue: Y = A^T M A per (tile, channel).  Accumulator register r is tile (r&3) + 8*(r>>2) + 4*(lane>>5), the lane's column is the
    // output channel.  The row stage (over xi) is local; the column stage sums over nu, half of which the partner wave (same tiles and channels,
    // other nu half) holds: each wave sends its partial sums of the two output columns the partner finalises through LDS (the operand images
    // are dead: the K loop ended with a barrier) and finalises its own two -- bias, BatchNorm partial sums, activation -- into the same staging
    // image, which then leaves as row-contiguous 16-byte stores.  The staging image is [row][column pair][channel][2]: the two columns a wave
    // sends (or finalises) for a tile row are one 8-byte slot per lane (32 lanes = 64 banks).  Every loop over r is unrolled by construction
    // (static_for: a rolled loop indexes the accumulators through s_set_gpr_idx and took 39 k cycles per block) and the nu half is a template
    // argument of the stage (a run-time half computed both column stages and selected).
    const int co = ct * XC + wn * 32 + (lane & 31);
    const bool cval = co < p.Co;
    const float bv = (p.flags & PC_F_BIAS) && cval ? p.bias[co] : 0.f;
    const bool accum = p.flags & PC_F_ACCUM;
    float s1 = 0.f, s2 = 0.f;
    const size_t plane_out = (size_t)p.H * p.W * p.ldo;
    const bool vec_ok = !(VAR & (8 | 32)) && p.ldo % 4 == 0 && p.Co % 4 == 0 && ((uintptr_t)p.out % 16 == 0);
    float* Tst = smem;                                    // [4 BTH rows][2 BTW column pairs][XC channels][2]
    const int OP2 = 2 * p.BTW, rowf = OP2 * XC * 2;        // column pairs per row / floats per row
    const int ntile = p.BTH * p.BTW;
    const int mhi = 4 * (lane >> 5);
    const int tcol = (wn * 32 + (lane & 31)) * 2;
    f32x2 own[16][4];
    auto stage1 = [&](auto NH) {
        constexpr int nh_ = decltype(NH)::value;
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int m = (r & 3) + 8 * (r >> 2) + mhi;
            const int li = (m * p.btw_magic) >> 16, lj = m - li * p.BTW;
            float S[4][3];
#pragma unroll
            for (int nul = 0; nul < 3; ++nul) {
                const float m0 = acc[nul][r], m1 = acc[3 + nul][r], m2 = acc[6 + nul][r], m3 = acc[9 + nul][r], m4 = acc[12 + nul][r], m5 = acc[15 + nul][r];
                const float pp = m1 + m2, qq = m1 - m2, rr = m3 + m4, ss = m3 - m4;
                S[0][nul] = m0 + pp + rr;
                S[1][nul] = __builtin_fmaf(PB, ss, PA * qq);
                S[2][nul] = __builtin_fmaf(PB2, rr, PA2 * pp);
                S[3][nul] = __builtin_fmaf(PB3, ss, PA3 * qq) + m5;
            }
            // this wave sends the column pair 1 - nh and keeps pair nh
            float* ts = Tst + (4 * li * OP2 + 2 * lj + (1 - nh_)) * (XC * 2) + tcol;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 keep, send;
                if constexpr (nh_ == 0) {      // nu = 0, 1, 2: A^T columns (1 0 0 0), (1 a a2 a3), (1 -a a2 -a3)
                    const float P_ = S[i][1] + S[i][2], Q_ = S[i][1] - S[i][2];
                    keep = (f32x2){S[i][0] + P_, PA * Q_};
                    send = (f32x2){PA2 * P_, PA3 * Q_};
                } else {                       // nu = 3, 4, 5: A^T columns (1 b b2 b3), (1 -b b2 -b3), (0 0 0 1)
                    const float P_ = S[i][0] + S[i][1], Q_ = S[i][0] - S[i][1];
                    send = (f32x2){P_, PB * Q_};
                    keep = (f32x2){PB2 * P_, __builtin_fmaf(PB3, Q_, S[i][2])};
                }
                own[r][i] = keep;
                if (m < ntile) *(f32x2*)(ts + i * rowf) = send;
            }
        });
    };
    if (nh == 0) stage1(std::integral_constant<int, 0>{});
    else stage1(std::integral_constant<int, 1>{});
    if (VAR & 64) stamp[3] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const int jown = 2 * nh;
    float* obase = p.out + ((size_t)n * p.T + t) * plane_out + co;
    auto finalise = [&](auto VEC) {
        constexpr bool vec = decltype(VEC)::value;
        const bool relu = p.act == PC_ACT_RELU;
        static_for<16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int m = (r & 3) + 8 * (r >> 2) + mhi;
            const int li = (m * p.btw_magic) >> 16, lj = m - li * p.BTW;
            const int oi = bh * p.BTH + li, oj = bw * p.BTW + lj;
            const bool ok = cval && m < ntile && oi < p.TH && oj < p.TW;
            float* ts = Tst + (4 * li * OP2 + 2 * lj + nh) * (XC * 2) + tcol;
            if (ok) {
                f32x2 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *(const f32x2*)(ts + i * rowf);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] += own[r][i] + (f32x2){bv, bv};
                    s1 += v[i][0] + v[i][1]; s2 += v[i][0] * v[i][0] + v[i][1] * v[i][1];
                    if (relu) { v[i][0] = fmaxf(v[i][0], 0.f); v[i][1] = fmaxf(v[i][1], 0.f); }
                }
                if constexpr (vec) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) *(f32x2*)(ts + i * rowf) = v[i];
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float vv = v[q >> 1][q & 1];
                        float* o = obase + ((size_t)(4 * oi + (q >> 1)) * p.W + 4 * oj + jown + (q & 1)) * p.ldo;
                        if (accum) vv += *o;
                        if (!(VAR & 8) || vv == 12345.678f) *o = vv;
                    }
                }
            }
        });
    };
    if (vec_ok) finalise(std::true_type{});
    else finalise(std::false_type{});
    if (VAR & 64) stamp[4] = __builtin_amdgcn_s_memtime();
    if (vec_ok) {
        __syncthreads();
        const int c4 = tid & 15, co0 = ct * XC + c4 * 4;
        const int npp = 8 * ntile;                                       // column pairs of the block
        float* ob = p.out + ((size_t)n * p.T + t) * plane_out + co0;
        if (co0 < p.Co) {
            int lr = (tid >> 4) / OP2, lcp = (tid >> 4) - lr * OP2;      // this thread's column pair: every 16th, walked without a division
#pragma unroll 2
            for (int pp = tid >> 4; pp < npp; pp += 16) {
                const int orow = 4 * bh * p.BTH + lr, ocol = 4 * bw * p.BTW + 2 * lcp;
                if (orow < 4 * p.TH && ocol < 4 * p.TW) {
                    const float* tp = Tst + (pp * XC + c4 * 4) * 2;
                    const f32x4 a = *(const f32x4*)tp, b = *(const f32x4*)(tp + 4);          // (c0 p0, c0 p1, c1 p0, c1 p1), (c2 .., c3 ..)
                    f32x4 v0 = {a[0], a[2], b[0], b[2]}, v1 = {a[1], a[3], b[1], b[3]};
                    float* o = ob + ((size_t)orow * p.W + ocol) * p.ldo;
                    if (accum) { v0 += *(const f32x4*)o; v1 += *(const f32x4*)(o + p.ldo); }
                    *(f32x4*)o = v0;
                    *(f32x4*)(o + p.ldo) = v1;
                }
                lcp += 16;
                while (lcp >= OP2) { lcp -= OP2; ++lr; }
            }
        }
    }
    if (VAR & 64) {
        stamp[5] = __builtin_amdgcn_s_memtime();
        if (tid == 0 && (p.flags & PC_F_BNPART) == 0 && p.bnpart) {
            unsigned long long* dbg = (unsigned long long*)p.bnpart + (size_t)blockIdx.x * 8;
            dbg[0] = stamp[1] - stamp[0]; dbg[1] = stamp[2] - stamp[1]; dbg[2] = stamp[3] - stamp[2]; dbg[3] = stamp[4] - stamp[3]; dbg[4] = stamp[5] - stamp[4];
            dbg[5] = (unsigned long long)nchunks; dbg[6] = 0; dbg[7] = 0;
        }
    }
    if (VAR & 32) {
        stamp[3] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned long long* dbg = (unsigned long long*)p.bnpart + (size_t)blockIdx.x * 4;
            dbg[0] = stamp[1] - stamp[0]; dbg[1] = stamp[2] - stamp[1]; dbg[2] = stamp[3] - stamp[2]; dbg[3] = (unsigned long long)nchunks;
        }
        return;
    }
    if (p.flags & PC_F_BNPART) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float* part = p.bnpart + ((size_t)sblock * 2 + nh) * 2 * p.Co;
        if (lane < 32 && cval) { part[co] = s1; part[p.Co + co] = s2; }
    }
}

// Block rectangle: BTH x BTW <= 32 tiles, fewest wasted tile slots, then the smallest raw patch; the patch image must fit its LDS-DMA pieces
void choose_block(int TH, int TW, int& bth, int& btw) {
    double best = 1e30;
    bth = 4; btw = 8;
    for (int w = 1; w <= XT && w <= TW; ++w) {
        int h = XT / w;
        if (h > TH) h = TH;
        if (h < 1) continue;
        if ((4 * h + 2) * (4 * (w + 1)) > RPIECES * 64) continue;
        const double blocks = (double)cdiv(TH, h) * cdiv(TW, w);
        const double waste = blocks * XT / ((double)TH * TW);
        const double aspect = (double)(4 * w + 2) * (4 * h + 2) / (16.0 * w * h);
        const double cost = waste * (1.0 + 0.05 * aspect);
        if (cost < best - 1e-9) { best = cost; bth = h; btw = w; }
    }
}

// Row pitch of the raw-patch image: the transform's ds_read_b128 of patch position (4 ti + r, 4 tj + c) is issued by lane groups of 16 tiles
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) and a group is conflict-free when its tiles sit on 16 different
// 16-byte slots modulo 16.  Tiles of one tile row are consecutive slots; tile rows are 4 rpitch apart.
void choose_pitch(int bth, int btw, int& rpitch, int& rq) {
    const int PH = 4 * bth + 2;
    rq = btw + 1;
    static const int lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int best = 1 << 30;
    rpitch = 4 * rq;
    for (int pitch = 4 * rq; pitch <= 4 * rq + 15 && PH * pitch <= RPIECES * 64; ++pitch) {
        int cost = 0;
        for (int gset = 0; gset < 2; ++gset) {
            int cnt[16] = {0}, mx = 0;
            for (int q = 0; q < 16; ++q) {
                const int tile = lanes[gset][q];
                if (tile >= bth * btw) continue;
                const int ti = tile / btw, tj = tile - ti * btw;
                const int c = ++cnt[(4 * ti * pitch + tj) & 15];
                mx = c > mx ? c : mx;
            }
            cost += mx > 1 ? mx - 1 : 0;
        }
        if (cost < best) { best = cost; rpitch = pitch; }
    }
}

int fill(const pc_wino_desc* d, Wino4K& k) {
    PC_CHECK_ARG(d, "pc_wino: null descriptor");
    PC_CHECK_ARG(d->N >= 1 && d->T >= 1 && d->H >= 4 && d->W >= 4 && d->H % 4 == 0 && d->W % 4 == 0, "pc_wino (m = 4): H, W must be multiples of 4 (H=%d W=%d)", d->H, d->W);
    PC_CHECK_ARG(d->Ci >= 8 && d->Ci % 8 == 0 && d->ldi % 4 == 0 && d->ldi >= d->Ci && d->Co >= 1 && d->ldo >= d->Co,
                 "pc_wino (m = 4): Ci %% 8, ldi %% 4, ldi >= Ci, ldo >= Co (Ci=%d ldi=%d Co=%d ldo=%d)", d->Ci, d->ldi, d->Co, d->ldo);
    PC_CHECK_ARG(d->KT == 1 || d->KT == 3, "pc_wino: KT must be 1 or 3");
    PC_CHECK_ARG((int64_t)d->N * (d->T > d->Ti ? d->T : d->Ti) * d->H * d->W * (int64_t)(d->ldi > d->ldo ? d->ldi : d->ldo) < (1ll << 40), "pc_wino: tensor too large");
    PC_CHECK_ARG((int64_t)d->H * d->W * d->ldi * 4 < 0xff000000ll, "pc_wino: plane too large (the LDS-DMA lane offsets are 32-bit byte offsets inside one frame)");
    k.N = d->N; k.T = d->T; k.H = d->H; k.W = d->W; k.Ci = d->Ci; k.ldi = d->ldi; k.Co = d->Co; k.ldo = d->ldo;
    k.TH = d->H / 4; k.TW = d->W / 4;
    choose_block(k.TH, k.TW, k.BTH, k.BTW);
    choose_pitch(k.BTH, k.BTW, k.rpitch, k.rq);
    PC_CHECK_ARG((4 * k.BTH + 2) * k.rpitch <= RPIECES * 64, "pc_wino (m = 4): no block rectangle fits H=%d W=%d", d->H, d->W);
    k.nbh = cdiv(k.TH, k.BTH); k.nbw = cdiv(k.TW, k.BTW);
    k.btw_magic = (65536 + k.BTW - 1) / k.BTW;
    k.nct = cdiv(d->Co, XC); k.nc4 = d->Ci / XK;
    k.KT = d->KT; k.act = d->act; k.flags = d->flags;
    PC_CHECK_ARG(d->Ti >= 1 && d->ta >= 1 && d->tden >= 1, "pc_wino: Ti / ta / tden must be >= 1 (Ti=%d ta=%d tden=%d)", d->Ti, d->ta, d->tden);
    k.Ti = d->Ti; k.ta = d->ta; k.tc = d->tc; k.tden = d->tden;
    return PC_OK;
}

}  // namespace

extern "C" int64_t pc_wino4_u_floats(int O, int I, int KT) {
    if (O < 1 || I < 8 || I % 8 || (KT != 1 && KT != 3)) return -1;
    return (int64_t)KT * cdiv(O, XC) * (I / XK) * UPLANE;
}

extern "C" int pc_wino4_weights(const float* w, int64_t sO, int64_t sT, int64_t sI, int O, int I, int KT, int flip, float* U, pc_stream s) {
    PC_CHECK_ARG(w && U, "pc_wino4_weights: null pointer");
    PC_CHECK_ARG(O >= 1 && I >= 8 && I % 8 == 0 && (KT == 1 || KT == 3), "pc_wino4_weights: O=%d I=%d KT=%d", O, I, KT);
    const int nct = cdiv(O, XC), nc4 = I / XK;
    const long long total = (long long)KT * nct * 64 * I;
    hipLaunchKernelGGL(wino4_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, (long long)sO, (long long)sT, (long long)sI, O, I, KT, flip, U, nct, nc4);
    PC_CHECK_LAUNCH("wino4_weights_kernel");
    return PC_OK;
}

// The m = 4 halves of pc_wino_bnpart_rows / pc_wino_work / pc_wino_conv (wino.hip dispatches on pc_wino_desc.m)
int pc_wino4_bnpart_rows_impl(const pc_wino_desc* d) {
    Wino4K k;
    if (fill(d, k) != PC_OK) return -1;
    return k.N * k.T * k.nbh * k.nbw * 2;
}

int pc_wino4_work_impl(const pc_wino_desc* d, double* out) {
    Wino4K k;
    const int rc = fill(d, k);
    if (rc != PC_OK) return rc;
    PC_CHECK_ARG(out, "pc_wino_work: null pointer");
    double taps = 0;
    for (int t = 0; t < k.T; ++t)
        for (int a = 0; a < k.KT; ++a) { const int num = t * k.ta + a + k.tc; taps += num >= 0 && num % k.tden == 0 && num / k.tden < k.Ti; }
    const double blocks = (double)k.N * k.nbh * k.nbw * k.nct;
    out[0] = blocks * taps * 36.0 * XT * XC * k.Ci;                                  // issued: 36 transform-domain GEMMs of 32 x 64 x Ci per tap
    out[1] = (double)k.N * taps * 36.0 * ((double)k.TH * k.TW) * k.Co * k.Ci;        // executed on real tiles / channels
    out[2] = blocks * k.T;
    return PC_OK;
}

int pc_wino4_conv_impl(const pc_wino_desc* d, const float* in, const float* U, const float* bias, float* out, float* bnpart, pc_stream s) {
    Wino4K k;
    const int rc = fill(d, k);
    if (rc != PC_OK) return rc;
    PC_CHECK_ARG(in && U && out, "pc_wino_conv: null pointer");
    PC_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)U % 16 == 0), "pc_wino_conv: in / U must be 16-byte aligned");
    PC_CHECK_ARG(!(d->flags & PC_F_BIAS) || bias, "pc_wino_conv: bias flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_BNPART) || bnpart, "pc_wino_conv: bnpart flag without pointer");
    PC_CHECK_ARG(!(d->flags & ~(PC_F_BIAS | PC_F_ACCUM | PC_F_BNPART)), "pc_wino_conv: unsupported flag");
    PC_CHECK_ARG(d->act == PC_ACT_NONE || d->act == PC_ACT_RELU, "pc_wino_conv: activation");
    k.in = in; k.U = U; k.bias = bias; k.out = out; k.bnpart = bnpart;
#ifdef PICONS_DIAG
    static const int var = getenv("PICONS_WINO_VARIANT") ? atoi(getenv("PICONS_WINO_VARIANT")) : 0;
    PC_CHECK_ARG(var != 32 || bnpart, "pc_wino_conv: variant 32 writes its stamps through bnpart");
#endif
    const size_t lds = (size_t)(16 * XT * XC) * sizeof(float);        // the epilogue's staging image (the K loop's R and V images are 60 KiB of it)
    static_assert(2 * RPLANE + 2 * VPLANE <= 16 * XT * XC, "the operand images must fit the staging image");
    const dim3 grid((unsigned)((int64_t)k.N * k.T * k.nbh * k.nbw * k.nct));
#define WINO_LAUNCH(V)                                                                                                            \
    {                                                                                                                             \
        PC_SET_LDS_ONCE(wino4_conv_kernel<V>, lds, "wino4_conv_kernel");                                                          \
        if (pc_tl_ev_start) hipExtLaunchKernelGGL(wino4_conv_kernel<V>, grid, dim3(256), lds, (hipStream_t)s, pc_tl_ev_start, pc_tl_ev_stop, 0, k); \
        else hipLaunchKernelGGL(wino4_conv_kernel<V>, grid, dim3(256), lds, (hipStream_t)s, k);                                  \
    }
#ifdef PICONS_DIAG
    switch (var) {
        case 1: WINO_LAUNCH(1) break;
        case 2: WINO_LAUNCH(2) break;
        case 3: WINO_LAUNCH(3) break;
        case 4: WINO_LAUNCH(4) break;
        case 7: WINO_LAUNCH(7) break;
        case 8: WINO_LAUNCH(8) break;
        case 15: WINO_LAUNCH(15) break;
        case 32: WINO_LAUNCH(32) break;
        case 64: WINO_LAUNCH(64) break;
        default: WINO_LAUNCH(0) break;
    }
#else
    WINO_LAUNCH(0)
#endif
#undef WINO_LAUNCH
    PC_CHECK_LAUNCH("wino4_conv_kernel");
    return PC_OK;
}
