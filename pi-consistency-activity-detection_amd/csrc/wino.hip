// Winograd F(2x2, 3x3) over (h, w) for the stride-1, pad-1 3x3x3 / 1x3x3 convolutions that dominate the step (the decoder's
// skip convs conv112 / conv56, capsules_ucf101.py:382-384,497,501; Conv3d_2c, pytorch_i3d.py:236-238; and their input
// gradients, which are the same correlation with mirrored taps): 2.25x fewer multiply-accumulates than the gather-GEMM form,
// the temporal taps stay direct.  ONE fused kernel: the 4x4 input patches are transformed in registers on their way into LDS,
// the 16 transform-domain GEMMs run on v_mfma_f32_32x32x2_f32 with all 16 accumulators of a (32 tiles x 32 channels) sub-tile
// in one wave's registers, and the inverse transform is register-local in the epilogue -- the transform-domain tensors (4x the
// activations) never exist in HBM.
//
//   Y = A^T [ sum_ci sum_kt (G g_kt G^T) .* (B^T d_kt B) ] A       per 2x2 output tile, d = 4x4 input patch at (2i-1, 2j-1)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// 1 KiB LDS-DMA piece through a raw buffer resource: wave-uniform base + one 32-bit byte offset per lane; a lane whose offset is DMA_OOB gets
// zeros (see conv.hip, glds16b: this form costs the matrix pipe less than half of what 64-bit per-lane addresses cost).  In this kernel every
// lane offset is a per-block constant -- the K loop spends no vector instruction on DMA addresses.
constexpr unsigned DMA_OOB = 0xffffffffu;
typedef __amdgpu_buffer_rsrc_t dma_rsrc_t;
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffff00u, 0x00020000);
}
template <bool ON = true>
__device__ __forceinline__ void glds16b(dma_rsrc_t rs, unsigned voff, float* l) {
    if (ON) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}

struct WinoK {
    const float* in; const float* U; const float* bias; float* out; float* bnpart;
    int N, T, H, W, Ci, ldi, Co, ldo;
    int TH, TW, BTH, BTW, nbh, nbw, nct, nc8;
    int scalar_epi;                         // PICONS_WINO_SCALAR_EPI=1: four-byte stores straight from the accumulators (A/B switch)
    int KT, act, flags;
    int Ti, ta, tc, tden;
    int btw_magic;
    int rpitch, rhalf;                      // raw-patch image: positions per patch row / offset of the odd columns inside a row (choose_pitch)
    int strip, nsp;                         // strip mode (below): blocks per pair of planes = strips per plane
};

constexpr int WT = 64;            // tiles per block (rows of the transform-domain GEMMs)
constexpr int WC = 64;            // output channels per block
constexpr int WK = 8;             // input channels per K chunk
constexpr int PLANE = 16 * 2 * 64 * 4;     // floats of one LDS operand image: [xi*4+nu][k half][row 64][4]

// ---- weight transform: U[kt][ct][c8][xi*4+nu][kh][co 64][4] = (G g G^T)[xi][nu] of g = w[o][kt][.][.][i], read through strides so
// the master OIDHW tensor (forward) and its transpose with mirrored taps (input gradient) need no intermediate layout.
__global__ void wino_weights_kernel(const float* __restrict__ w, long long sO, long long sT, long long sI, int O, int I, int KT, int flip,
                                    float* __restrict__ U, int nct, int nc8) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)KT * nct * 64 * I;
    if (e >= total) return;
    const int i = (int)(e % I);
    long long r = e / I;
    const int o = (int)(r % (nct * 64));
    const int kt = (int)(r / (nct * 64));
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ks = flip ? KT - 1 - kt : kt, as = flip ? 2 - a : a, bs = flip ? 2 - b : b;
            g[a][b] = o < O ? w[(long long)o * sO + (long long)((ks * 3 + as) * 3 + bs) * sT + (long long)i * sI] : 0.f;
        }
    float t[4][3];          // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    float* dst = U + ((((long long)kt * nct + o / 64) * nc8 + i / 8) * 16) * (2 * 64 * 4) + ((long long)((i & 7) >> 2) * 64 + (o & 63)) * 4 + (i & 3);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float u0 = t[x][0], u1 = 0.5f * (t[x][0] + t[x][1] + t[x][2]), u2 = 0.5f * (t[x][0] - t[x][1] + t[x][2]), u3 = t[x][2];
        dst[(x * 4 + 0) * (2 * 64 * 4)] = u0;
        dst[(x * 4 + 1) * (2 * 64 * 4)] = u1;
        dst[(x * 4 + 2) * (2 * 64 * 4)] = u2;
        dst[(x * 4 + 3) * (2 * 64 * 4)] = u3;
    }
}

// ---- the fused convolution.  Block = 64 tiles (a BTH x BTW rectangle of 2x2-output tiles of one (n, t) plane) x 64 output channels,
// four waves as 2 (tile halves) x 2 (channel halves), one wave per SIMD with all 16 transform-domain accumulators (256 registers).
// K chunk = 8 input channels of one temporal tap.  Three LDS images per chunk, all filled by LDS-DMA or by the block itself:
//   R  the raw (2 BTH + 2) x (2 BTW + 2) input patch of the block, [position][8 channels] (32 B per position, zero line for padding):
//      each input element is fetched ONCE per block (per-tile register loads fetched it ~4.5 times as 16-byte pieces of 64 different
//      cache lines per instruction, and cost a third of the kernel).  Round 5: the positions of a patch row are stored EVEN COLUMNS FIRST
//      (position (r, c) at r * rpitch + (c & 1) * rhalf + c / 2) and a lane of the transform is (tile lane / 2, channel half lane % 2): the
//      16 lanes of a ds_read_b128 group then read 8 tiles x 2 halves = 16 consecutive 16-byte slots of one parity block instead of 16
//      slots 64 bytes apart (4-way conflicts: SQ_LDS_BANK_CONFLICT was 37 % of the kernel's LDS-active cycles in rounds 3 and 4); rpitch is
//      chosen on the host so that the tiles of a group that sit in different tile rows do not meet either (choose_pitch),
//   V  the transformed patch B^T d B, [xi*4+nu][k half][tile ^ 4 (k half)][4], written by the block: thread = (tile, 4-channel half, two of the
//      four B^T rows): 12 ds_read_b128 of R, 16 float4 adds, 8 ds_write_b128 (the XOR keeps the two halves of a store group on different banks),
//   U  the transformed weights, contiguous in HBM by construction.
// R runs two chunks ahead (its DMA has a whole chunk to land), U one; one barrier per chunk.
// STRIP MODE (round 6; frames whose tile grid is a multiple of 14 wide: the 28 x 28 and 56 x 56 layers).  A 14 x 14-tile frame in 64-tile
// rectangles is four blocks of 49 tiles: 49 of 64 tile slots multiply real data.  Instead a block is TWO strips of 2 x 14 tiles (56 of 64
// slots), each with its own 6 x 30 raw patch, taken from the strips of a PAIR of planes (n, n + 1) of the same t in order: seven blocks per two
// 28 x 28 frames instead of eight; only the middle block of a pair has its strips in two different planes (same t, so the same temporal taps
// and weight chunks).  Everything behind the patch fetch and in front of the output store is unchanged: tile (row li of 4, column lj of 14)
// of the block reads patch rows 6 (li / 2) + 2 (li % 2) .. + 3.
// VAR (diagnostics, PICONS_WINO_VARIANT; results are WRONG with bits 1-8): 1 = no patch DMA, 2 = no U DMA, 4 = no transform stores,
// 8 = no output stores, 32 = s_memtime stamps of prologue / K loop / epilogue per block into bnpart (4 x u64 per block).
constexpr int RMAX = 12;          // 1 KiB DMA pieces of one raw patch image (three per wave): up to 384 positions (an 8 x 8 tile block has 18 x 18 = 324)
constexpr int RPLANE = RMAX * 256;

template <int VAR, bool STRIP = false>
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(const WinoK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                  // [2][PLANE]
    float* Us = smem + 2 * PLANE;      // [2][PLANE]
    float* Rs = smem + 4 * PLANE;      // [2][RPLANE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keeps LDS bases / role selects in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    int sb = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = sb % p.nct; sb /= p.nct;
    const int sblock = sb;             // spatial block id (n, t, bh, bw): the BatchNorm partial row
    const int bw = sb % p.nbw; sb /= p.nbw;
    int bh = 0, t, n, dn = 0, tia = 0, tib = 0;        // strip mode: tile row of the first / second strip in its plane, second strip's plane = n + dn
    if constexpr (STRIP) {
        const int sp = sb % p.nsp; sb /= p.nsp;
        t = sb % p.T;
        const int ga = 2 * sp, gb = ga + 1, pa = ga >= p.nsp ? 1 : 0, pb = gb >= p.nsp ? 1 : 0;    // strips ga, gb of the pair's 2 nsp; nsp strips per plane
        n = 2 * (sb / p.T) + pa; dn = pb - pa;
        tia = 2 * (ga - pa * p.nsp); tib = 2 * (gb - pb * p.nsp);
    } else {
        bh = sb % p.nbh; sb /= p.nbh;
        t = sb % p.T; n = sb / p.T;
    }
    const int PW = 2 * p.BTW + 2, PH = STRIP ? 12 : 2 * p.BTH + 2, npos = PH * PW;
    const int h0 = 2 * bh * p.BTH - 1, w0 = 2 * bw * p.BTW - 1;          // image position of patch position (0, 0)
    const unsigned dn_in = (unsigned)((size_t)dn * p.Ti * p.H * p.W * p.ldi * 4);      // byte distance of the second strip's plane (same t, next n)

    // raw-patch DMA role: piece j of this wave is piece wave + 4 j of the image; lane -> (position q = piece * 32 + lane / 2, 4-channel half)
    unsigned roff[3];                  // byte offset inside the source plane, DMA_OOB for padding positions (the DMA writes zeros there)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = (wave + 4 * j) * 32 + (lane >> 1);                 // slot of the image: row q / rpitch, even columns first
        const int pr = q / p.rpitch, rem = q - pr * p.rpitch;
        const int odd = rem >= p.rhalf ? 1 : 0, idx = rem - odd * p.rhalf;
        const int pc = 2 * idx + odd;
        const int second = (STRIP && pr >= 6) ? 1 : 0;                   // patch rows 6 .. 11: the second strip's 6 x 30 patch
        const int h = STRIP ? 2 * (second ? tib : tia) - 1 + (pr - 6 * second) : h0 + pr, w = w0 + pc;
        const bool ok = pr < PH && idx < p.BTW + 1 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        roff[j] = ok ? (unsigned)(((h * p.W + w) * p.ldi + (lane & 1) * 4) * 4) + (second ? dn_in : 0u) : DMA_OOB;
    }
    const unsigned uoff = lane * 16;   // U pieces are contiguous images

    // transform role: thread = (tile, k half, row pair); reads 3 rows x 4 columns of R.  Lane = (tile lane / 2 of the wave's 32, k half lane % 2)
    const int ttile = (wave & 1) * 32 + (lane >> 1), tkh = lane & 1, thalf = wave >> 1;
    const int tli = ttile / p.BTW, tlj = ttile - tli * p.BTW;
    const bool tval = ttile < p.BTH * p.BTW;       // (tiles beyond the image read zero lines: their patch positions are padding)
    const int rbase = tval ? (((2 * tli + thalf + (STRIP ? 2 * (tli >> 1) : 0)) * p.rpitch + tlj) * 2 + tkh) * 4 : tkh * 4;      // float offset of patch (row thalf, col 0) in R
    // temporal taps whose source frame exists: tap kt reads input frame (t * ta + kt + tc) / tden when that is an integer in [0, Ti)
    int nkt = 0, ktl0 = 0, ktl1 = 0, ktl2 = 0, ttl0 = 0, ttl1 = 0, ttl2 = 0;
    for (int kt = 0; kt < p.KT; ++kt) {
        const int num = t * p.ta + kt + p.tc;
        if (num < 0 || num % p.tden) continue;
        const int tt = num / p.tden;
        if (tt >= p.Ti) continue;
        if (nkt == 0) { ktl0 = kt; ttl0 = tt; } else if (nkt == 1) { ktl1 = kt; ttl1 = tt; } else { ktl2 = kt; ttl2 = tt; }
        ++nkt;
    }
    const int nchunks = nkt * p.nc8;
    const size_t plane_in = (size_t)p.H * p.W * p.ldi;
    // chunk c = (valid tap q, 8-channel slice c8).  Two chunk counters run ahead of the MFMA loop -- the U stream one chunk, the raw-patch
    // stream two -- and are advanced incrementally (an integer division per chunk and stream cost 20 % of the kernel); both stop at the
    // last chunk, which the tail iterations re-fetch harmlessly.
    const float* rtap0 = p.in + ((size_t)n * p.Ti + ttl0) * plane_in;
    const float* rtap1 = p.in + ((size_t)n * p.Ti + ttl1) * plane_in;
    const float* rtap2 = p.in + ((size_t)n * p.Ti + ttl2) * plane_in;
    const float* utap0 = p.U + (((size_t)ktl0 * p.nct + ct) * p.nc8) * PLANE + wave * 8 * 256;
    const float* utap1 = p.U + (((size_t)ktl1 * p.nct + ct) * p.nc8) * PLANE + wave * 8 * 256;
    const float* utap2 = p.U + (((size_t)ktl2 * p.nct + ct) * p.nc8) * PLANE + wave * 8 * 256;
    auto advance = [&](int& q, int& c8) {
        const bool wrap = c8 + 1 == p.nc8, last = wrap && q + 1 >= nkt;
        c8 = last ? c8 : (wrap ? 0 : c8 + 1);
        q = (wrap && !last) ? q + 1 : q;
    };
    auto r_base = [&](int q, int c8) -> const float* { return (q == 0 ? rtap0 : (q == 1 ? rtap1 : rtap2)) + c8 * WK; };
    auto u_base = [&](int q, int c8) -> const float* { return (q == 0 ? utap0 : (q == 1 ? utap1 : utap2)) + (size_t)c8 * PLANE; };
    int uq = 0, uc8 = 0, rq = 0, rc8 = 0;

    const float* rnext = nullptr;                 // source plane / channel slice of the chunk whose raw patch the loop fetches
    auto issue_r = [&](const float* base, int rb) {           // raw patch -> R[rb] (prologue)
        float* rl = Rs + rb * RPLANE + wave * 256;
        const dma_rsrc_t rs = dma_rsrc(base);
#pragma unroll
        for (int j = 0; j < 3; ++j) glds16b<!(VAR & 1)>(rs, roff[j], rl + j * 4 * 256);
    };
    const float* ug = nullptr;
    float* ul = nullptr;
    f32x4 d[12], x0[4], x1[4];
    const float* rsrc = nullptr;
    float* vb = nullptr;
    auto read2 = [&](int g, int c0) {             // two columns of patch row g of this thread's three rows (tiles beyond the block's
#pragma unroll                                    // rectangle read some valid position: their rows are never stored)
        for (int c = c0; c < c0 + 2; ++c) d[g * 4 + c] = *(const f32x4*)(rsrc + (g * p.rpitch + (c & 1) * p.rhalf + (c >> 1)) * 8);
    };
    // B^T rows held by this thread: half 0 -> rows 0, 1 (d0 - d2, d1 + d2) of patch rows (d0, d1, d2); half 1 -> rows 3, 2 (d1 - d3, d2 - d1)
    // of patch rows (d1, d2, d3).  q = 0: first of the pair; q = 1: second, as d[1] + ca * d[0] + cb * d[2] with wave-uniform (ca, cb) =
    // (0, 1) / (-1, 0): no branch inside the K loop (a branch splits the loop into blocks and every s_waitcnt degrades to (0))
    const float ca = thalf ? -1.f : 0.f, cb = thalf ? 0.f : 1.f;
    auto xcol = [&](int q, int c) {
        if (q == 0) x0[c] = d[c] - d[8 + c];
        else x1[c] = d[4 + c] + ca * d[c] + cb * d[8 + c];
    };
    auto store_out = [&](int q, int nu) {
        const int xi = thalf ? (q == 0 ? 3 : 2) : q;
        const f32x4* x = q == 0 ? x0 : x1;
        f32x4 v;
        if (nu == 0) v = x[0] - x[2];
        else if (nu == 1) v = x[1] + x[2];
        else if (nu == 2) v = x[2] - x[1];
        else v = x[1] - x[3];
        if (!(VAR & 4)) *(f32x4*)(vb + (xi * 4 + nu) * 512) = v;
    };
    // The side work of one K chunk, one small piece per MFMA gap (step s = 4 * group + slot).  Slot 0 carries the LDS side work (two reads
    // of R or one transform-row store) and, behind it, the NEXT group's fragment reads -- so the s_waitcnt lgkmcnt(0) in front of a group's first
    // MFMA waits for nothing younger than those fragments --, slot 2 one LDS-DMA piece, slots 2 / 3 the transform arithmetic.
    auto side = [&](auto S, int buf) {
        constexpr int s_ = decltype(S)::value;
        constexpr int g = s_ >> 2, e = s_ & 3;
        if constexpr (e == 0 && g < 6) read2(g >> 1, (g & 1) * 2);                                     // patch rows of chunk c + 1 out of R
        if constexpr (e == 0 && g >= 8 && g < 12) store_out(0, g - 8);
        if constexpr (e == 0 && g >= 12) store_out(1, g - 12);
        if constexpr (e == 2 && g < 8) glds16b<!(VAR & 2)>(dma_rsrc(ug + g * 256), uoff, ul + g * 256);  // U piece g of chunk c + 1
        if constexpr (e == 2 && g >= 8 && g < 11)                                                     // R piece g - 8 of chunk c + 2
            glds16b<!(VAR & 1)>(dma_rsrc(rnext), roff[g - 8], Rs + buf * RPLANE + (wave + 4 * (g - 8)) * 256);
        if constexpr (g == 6 && e >= 2) { xcol(0, (e - 2) * 2); xcol(0, (e - 2) * 2 + 1); }
        if constexpr (g == 7 && e >= 2) { xcol(1, (e - 2) * 2); xcol(1, (e - 2) * 2 + 1); }
    };
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long stamp[4] = {0, 0, 0, 0};
    if (VAR & 32) stamp[0] = __builtin_amdgcn_s_memtime();

    const int kh = lane >> 5;
    const int aoff = (kh * 64 + ((wm * 32 + (lane & 31)) ^ (kh * 4))) * 4, boff = (kh * 64 + wn * 32 + (lane & 31)) * 4;
    const int voff = (tkh * 64 + (ttile ^ (tkh * 4))) * 4;             // this thread's slot of a V plane
    // prologue: R(0), R(1), U(0) in flight; V(0) from R(0)
    if (nchunks > 0) {
        issue_r(r_base(rq, rc8), 0);
        advance(rq, rc8);
        issue_r(r_base(rq, rc8), 1);
        advance(rq, rc8);                         // -> chunk 2
        ug = u_base(uq, uc8);
        ul = Us + wave * 8 * 256;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16b<!(VAR & 2)>(dma_rsrc(ug + j * 256), uoff, ul + j * 256);
        advance(uq, uc8);                         // -> chunk 1
    }
    PC_SYNC_DMA();
    if (nchunks > 0) {
        rsrc = Rs + rbase;
        vb = Vs + voff;
#pragma unroll
        for (int g = 0; g < 3; ++g) { read2(g, 0); read2(g, 2); }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) xcol(q, c);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) store_out(q, nu);
        }
    }
    __syncthreads();
    if (VAR & 32) stamp[1] = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        // during chunk c: U(c+1) -> U[buf^1], R(c+2) -> R[buf] (R[buf] held chunk c, transformed one iteration ago), V(c+1) from R[buf^1]
        ug = u_base(uq, uc8);
        ul = Us + (buf ^ 1) * PLANE + wave * 8 * 256;
        advance(uq, uc8);
        rsrc = Rs + (buf ^ 1) * RPLANE + rbase;
        vb = Vs + (buf ^ 1) * PLANE + voff;
        const float* va = Vs + buf * PLANE + aoff;
        const float* ub = Us + buf * PLANE + boff;
        f32x4 a0 = *(const f32x4*)va, b0 = *(const f32x4*)ub, a1, b1;
        rnext = r_base(rq, rc8);
        advance(rq, rc8);
        // one group = the four MFMAs of one transform-domain position, with one piece of side work pinned behind each of them
        // (sched_barrier: the compiler otherwise gathers the side work, waits for it at once and strands the matrix pipe)
#define WINO_STEP(I, E, A, B)                                                                                  \
            acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[E], B[E], acc[I], 0, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            side(std::integral_constant<int, 4 * (I) + (E)>{}, buf);                                           \
            __builtin_amdgcn_sched_barrier(0);
#define WINO_GROUP(I, A, B, AN, BN)                                                                            \
        {                                                                                                      \
            WINO_STEP(I, 0, A, B)                                                                              \
            if ((I) + 1 < 16) { AN = *(const f32x4*)(va + ((I) + 1) * 512); BN = *(const f32x4*)(ub + ((I) + 1) * 512); } \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            WINO_STEP(I, 1, A, B) WINO_STEP(I, 2, A, B) WINO_STEP(I, 3, A, B)                                  \
        }
        WINO_GROUP(0, a0, b0, a1, b1)   WINO_GROUP(1, a1, b1, a0, b0)   WINO_GROUP(2, a0, b0, a1, b1)   WINO_GROUP(3, a1, b1, a0, b0)
        WINO_GROUP(4, a0, b0, a1, b1)   WINO_GROUP(5, a1, b1, a0, b0)   WINO_GROUP(6, a0, b0, a1, b1)   WINO_GROUP(7, a1, b1, a0, b0)
        WINO_GROUP(8, a0, b0, a1, b1)   WINO_GROUP(9, a1, b1, a0, b0)   WINO_GROUP(10, a0, b0, a1, b1)  WINO_GROUP(11, a1, b1, a0, b0)
        WINO_GROUP(12, a0, b0, a1, b1)  WINO_GROUP(13, a1, b1, a0, b0)  WINO_GROUP(14, a0, b0, a1, b1)  WINO_GROUP(15, a1, b1, a0, b0)
#undef WINO_STEP
#undef WINO_GROUP
        PC_SYNC_DMA();
    }

    if (VAR & 32) stamp[2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue: Y = A^T M A per (tile, channel), register-local: accumulator register r is tile row (r&3) + 8*(r>>2) + 4*(lane>>5) of
    // this wave's 32 tiles, the lane's column is the output channel
    const int co = ct * WC + wn * 32 + (lane & 31);
    const bool cval = co < p.Co;
    const float bv = (p.flags & PC_F_BIAS) && cval ? p.bias[co] : 0.f;
    const bool accum = p.flags & PC_F_ACCUM;
    float s1 = 0.f, s2 = 0.f;
    const size_t plane_out = (size_t)p.H * p.W * p.ldo;
    // Output path (round 4): the block's 2 BTH x 2 BTW output positions x 64 channels go through LDS (the operand images are dead: the K
    // loop ended with a barrier) and leave as row-contiguous 16-byte stores, 16 per thread -- instead of 64 four-byte stores per thread, each
    // with its own 64-bit address (12 k cycles of a conv112 block's 130 k).  Needs 16-byte aligned channel slices AND whole 4-channel chunks
    // (Co % 4 == 0: a chunk that straddles Co would overwrite the neighbouring slice of a wider tensor); otherwise the scalar path.
    const bool vec_ok = !(VAR & (8 | 32)) && !p.scalar_epi && p.ldo % 4 == 0 && p.Co % 4 == 0 && ((uintptr_t)p.out % 16 == 0);
    float* Tst = smem;                                    // [2 BTH * 2 BTW positions][WC channels]
    const int OW2 = 2 * p.BTW;
    float* obase = p.out + ((size_t)n * p.T + t) * plane_out + co;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int li = (m * p.btw_magic) >> 16, lj = m - li * p.BTW;      // m / BTW for m < 64 (exact: magic = ceil(65536 / BTW))
        const int second = (STRIP && li >= 2) ? 1 : 0;
        const int oi = STRIP ? (second ? tib : tia) + (li & 1) : bh * p.BTH + li, oj = bw * p.BTW + lj;
        const bool ok = cval && m < p.BTH * p.BTW && oi < p.TH && oj < p.TW;
        float s[4][2];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            s[x][0] = acc[x * 4 + 0][r] + acc[x * 4 + 1][r] + acc[x * 4 + 2][r];
            s[x][1] = acc[x * 4 + 1][r] - acc[x * 4 + 2][r] - acc[x * 4 + 3][r];
        }
        float y[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            y[0][b] = s[0][b] + s[1][b] + s[2][b];
            y[1][b] = s[1][b] - s[2][b] - s[3][b];
        }
        if (!ok) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v = y[a][b] + bv;
                s1 += v; s2 += v * v;
                if (p.act == PC_ACT_RELU) v = fmaxf(v, 0.f);
                if (vec_ok) { Tst[((2 * li + a) * OW2 + 2 * lj + b) * WC + wn * 32 + (lane & 31)] = v; continue; }
                float* o = obase + (second ? (size_t)dn * p.T * plane_out : 0) + ((size_t)(2 * oi + a) * p.W + 2 * oj + b) * p.ldo;
                if (accum) v += *o;
                if (!(VAR & 8) || v == 12345.678f) *o = v;
            }
    }
    if (vec_ok) {
        __syncthreads();
        const int c4 = tid & 15, co0 = ct * WC + c4 * 4;              // this thread's 16-byte channel chunk of every 16th position
        const int npo = 4 * p.BTH * p.BTW;
        float* ob = p.out + ((size_t)n * p.T + t) * plane_out + co0;
        if (co0 < p.Co) {
#pragma unroll 4
            for (int pos = tid >> 4; pos < npo; pos += 16) {
                const int lr = ((pos >> 1) * p.btw_magic) >> 16, lc = pos - lr * OW2;       // pos / (2 BTW)
                const int second = (STRIP && lr >= 4) ? 1 : 0;                              // output rows 4 .. 7 of the block: the second strip
                const int orow = STRIP ? 2 * (second ? tib : tia) + (lr & 3) : 2 * bh * p.BTH + lr, ocol = 2 * bw * p.BTW + lc;
                if (orow >= 2 * p.TH || ocol >= 2 * p.TW) continue;
                f32x4 v = *(const f32x4*)(Tst + pos * WC + c4 * 4);
                float* o = ob + (second ? (size_t)dn * p.T * plane_out : 0) + ((size_t)orow * p.W + ocol) * p.ldo;
                if (accum) v += *(const f32x4*)o;
                *(f32x4*)o = v;
            }
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
        float* part = p.bnpart + ((size_t)sblock * 2 + wm) * 2 * p.Co;
        if (lane < 32 && cval) { part[co] = s1; part[p.Co + co] = s2; }
    }
}

void choose_block(int TH, int TW, int& bth, int& btw) {
    double best = 1e30;
    bth = 8; btw = 8;
    for (int w = 1; w <= 64 && w <= TW; ++w) {
        int h = 64 / w;
        if (h > TH) h = TH;
        if (h < 1) continue;
        if ((2 * h + 2) * (2 * w + 2) * 2 > RMAX * 64) continue;          // the raw patch image must fit its LDS-DMA pieces (choose_pitch may pad its rows)
        const double blocks = (double)cdiv(TH, h) * cdiv(TW, w);
        const double waste = blocks * 64.0 / ((double)TH * TW);
        const double aspect = (double)(2 * w + 2) * (2 * h + 2) / (4.0 * w * h);      // patch read amplification: prefer square-ish
        const double cost = waste * (1.0 + 0.05 * aspect);
        if (cost < best - 1e-9) { best = cost; bth = h; btw = w; }
    }
}

// Row pitch of the raw-patch image (positions): the transform's ds_read_b128 of patch position (2 ti + r, 2 tj + c) is issued by lane groups
// of 16 = 8 tiles x 2 channel halves ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS), and a group is
// conflict-free when its 8 tiles sit on 8 different 32-byte positions modulo 8.  Tiles of one tile row are consecutive positions of a
// parity block; tiles of different rows are 2 rpitch apart per row -- pick the pitch (>= the patch width, image <= RMAX * 32 positions)
// with the fewest extra LDS cycles over all groups of both wave halves.
void choose_pitch(int bth, int btw, int& rpitch, int& rhalf, bool strip = false) {
    const int PW = 2 * btw + 2, PH = strip ? 12 : 2 * bth + 2;
    rhalf = btw + 1;
    static const int lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int best = 1 << 30;
    rpitch = PW;
    for (int pitch = PW; pitch <= PW + 15 && PH * pitch <= RMAX * 32; ++pitch) {
        int cost = 0;
        for (int half = 0; half < 4; ++half)            // (wave & 1, lanes 0-31 / 32-63): tiles 16 half ..
            for (int gset = 0; gset < 2; ++gset) {
                int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 0;
                for (int q = 0; q < 16; q += 2) {
                    const int tile = half * 16 + (lanes[gset][q] >> 1);
                    if (tile >= bth * btw) continue;
                    const int ti = tile / btw, tj = tile - ti * btw;
                    const int c = ++cnt[((2 * ti + (strip ? 2 * (ti >> 1) : 0)) * pitch + tj) & 7];
                    mx = c > mx ? c : mx;
                }
                cost += mx > 1 ? mx - 1 : 0;
            }
        if (cost < best) { best = cost; rpitch = pitch; }
    }
}

int fill(const pc_wino_desc* d, WinoK& k) {
    PC_CHECK_ARG(d, "pc_wino: null descriptor");
    PC_CHECK_ARG(d->N >= 1 && d->T >= 1 && d->H >= 2 && d->W >= 2 && d->H % 2 == 0 && d->W % 2 == 0, "pc_wino: H, W must be even (H=%d W=%d)", d->H, d->W);
    PC_CHECK_ARG(d->Ci >= 8 && d->Ci % 8 == 0 && d->ldi % 4 == 0 && d->ldi >= d->Ci && d->Co >= 1 && d->ldo >= d->Co,
                 "pc_wino: Ci %% 8, ldi %% 4, ldi >= Ci, ldo >= Co (Ci=%d ldi=%d Co=%d ldo=%d)", d->Ci, d->ldi, d->Co, d->ldo);
    PC_CHECK_ARG(d->KT == 1 || d->KT == 3, "pc_wino: KT must be 1 or 3");
    PC_CHECK_ARG((int64_t)d->N * (d->T > d->Ti ? d->T : d->Ti) * d->H * d->W * (int64_t)(d->ldi > d->ldo ? d->ldi : d->ldo) < (1ll << 40), "pc_wino: tensor too large");
    PC_CHECK_ARG((int64_t)d->H * d->W * d->ldi * 4 < 0xff000000ll, "pc_wino: plane too large (the LDS-DMA lane offsets are 32-bit byte offsets inside one frame)");
    k.N = d->N; k.T = d->T; k.H = d->H; k.W = d->W; k.Ci = d->Ci; k.ldi = d->ldi; k.Co = d->Co; k.ldo = d->ldo;
    k.TH = d->H / 2; k.TW = d->W / 2;
    choose_block(k.TH, k.TW, k.BTH, k.BTW);
    // strip mode (kernel header): tile grids a multiple of 14 wide and an even number of tile rows high, planes paired (n, n + 1) -- N % 4 == 0 keeps
    // a pair inside one BatchNorm batch group (groups <= 2: the partial rows of a block belong to one group)
    // Asked for per launch (PC_F_STRIPS); the planner leaves it OFF (measured, profiles/r06_wino_strips.txt): 12.5 % fewer blocks, but a strip block
    // fetches a 12 x 30 raw patch where a 7 x 7-tile rectangle fetches 16 x 16 (+40 %) and holds 56 instead of 49 tiles: the twenty 28 x 28 launches
    // take 1.83 ms single-stream against 1.60, the four-lane step is unchanged within noise.  Bit-identical results either way (tests/test_wino_gpu.py).
    const int strips = (d->flags & PC_F_STRIPS) ? 1 : 0;
    k.strip = strips && k.TW % 14 == 0 && k.TH % 2 == 0 && d->N % 4 == 0 && (int64_t)2 * d->Ti * d->H * d->W * d->ldi * 4 < 0xff000000ll &&
              (k.TH / 2) * (k.TW / 14) < 2 * cdiv(k.TH, k.BTH) * cdiv(k.TW, k.BTW);          // ... and only where strips are fewer blocks (56 x 56 tiles in 8 x 8 rectangles are exact)
    k.nsp = k.TH / 2;
    if (k.strip) { k.BTH = 4; k.BTW = 14; }
    choose_pitch(k.BTH, k.BTW, k.rpitch, k.rhalf, k.strip != 0);
    k.nbh = cdiv(k.TH, k.BTH); k.nbw = cdiv(k.TW, k.BTW);
    k.btw_magic = (65536 + k.BTW - 1) / k.BTW;
    k.nct = cdiv(d->Co, WC); k.nc8 = d->Ci / WK;
    static const int scalar_epi = getenv("PICONS_WINO_SCALAR_EPI") ? atoi(getenv("PICONS_WINO_SCALAR_EPI")) : 0;
    k.scalar_epi = scalar_epi;
    k.KT = d->KT; k.act = d->act; k.flags = d->flags;
    PC_CHECK_ARG(d->Ti >= 1 && d->ta >= 1 && d->tden >= 1, "pc_wino: Ti / ta / tden must be >= 1 (Ti=%d ta=%d tden=%d)", d->Ti, d->ta, d->tden);
    k.Ti = d->Ti; k.ta = d->ta; k.tc = d->tc; k.tden = d->tden;
    return PC_OK;
}

}  // namespace

// spatial blocks of a launch (x nct channel tiles = the grid): rectangles per plane, or -- strip mode -- nsp blocks per pair of planes
static inline int64_t wino_spatial_blocks(const WinoK& k) {
    return k.strip ? (int64_t)(k.N / 2) * k.T * k.nsp * k.nbw : (int64_t)k.N * k.T * k.nbh * k.nbw;
}

// pc_wino_desc.m == 4: the F(4x4, 3x3) kernel (wino4.hip)
int pc_wino4_bnpart_rows_impl(const pc_wino_desc* d);
int pc_wino4_work_impl(const pc_wino_desc* d, double* out);
int pc_wino4_conv_impl(const pc_wino_desc* d, const float* in, const float* U, const float* bias, float* out, float* bnpart, pc_stream s);
#define WINO_M_CHECK(d) PC_CHECK_ARG(!(d) || (d)->m == 0 || (d)->m == 2 || (d)->m == 4, "pc_wino: m must be 2 (or 0) or 4, got %d", (d)->m)

extern "C" int64_t pc_wino_u_floats(int O, int I, int KT) {
    if (O < 1 || I < 8 || I % 8 || (KT != 1 && KT != 3)) return -1;
    return (int64_t)KT * cdiv(O, WC) * (I / WK) * PLANE;
}

extern "C" int pc_wino_weights(const float* w, int64_t sO, int64_t sT, int64_t sI, int O, int I, int KT, int flip, float* U, pc_stream s) {
    PC_CHECK_ARG(w && U, "pc_wino_weights: null pointer");
    PC_CHECK_ARG(O >= 1 && I >= 8 && I % 8 == 0 && (KT == 1 || KT == 3), "pc_wino_weights: O=%d I=%d KT=%d", O, I, KT);
    const int nct = cdiv(O, WC), nc8 = I / WK;
    const long long total = (long long)KT * nct * 64 * I;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, (long long)sO, (long long)sT, (long long)sI, O, I, KT, flip, U, nct, nc8);
    PC_CHECK_LAUNCH("wino_weights_kernel");
    return PC_OK;
}

extern "C" int pc_wino_bnpart_rows(const pc_wino_desc* d) {
    if (d && d->m == 4) return pc_wino4_bnpart_rows_impl(d);
    if (d && d->m != 0 && d->m != 2) return -1;
    WinoK k;
    if (fill(d, k) != PC_OK) return -1;
    return wino_spatial_blocks(k) * 2;
}

// Host-only: multiply-accumulates the launch issues to the matrix cores / performs on real outputs (see pc_conv_work)
extern "C" int pc_wino_work(const pc_wino_desc* d, double* out) {
    WINO_M_CHECK(d);
    if (d && d->m == 4) return pc_wino4_work_impl(d, out);
    WinoK k;
    const int rc = fill(d, k);
    if (rc != PC_OK) return rc;
    PC_CHECK_ARG(out, "pc_wino_work: null pointer");
    double taps = 0;                                       // valid temporal taps summed over t
    for (int t = 0; t < k.T; ++t)
        for (int a = 0; a < k.KT; ++a) { const int num = t * k.ta + a + k.tc; taps += num >= 0 && num % k.tden == 0 && num / k.tden < k.Ti; }
    const double blocks = (double)wino_spatial_blocks(k) / k.T * k.nct;
    out[0] = blocks * taps * 16.0 * WT * WC * k.Ci;                                  // issued: 16 transform-domain GEMMs of 64 x 64 x Ci per tap
    out[1] = (double)k.N * taps * 16.0 * ((double)k.TH * k.TW) * k.Co * k.Ci;        // executed on real tiles / channels
    out[2] = blocks * k.T;                                                          // blocks
    return PC_OK;
}

extern "C" int pc_wino_conv(const pc_wino_desc* d, const float* in, const float* U, const float* bias, float* out, float* bnpart, pc_stream s) {
    WINO_M_CHECK(d);
    if (d && d->m == 4) return pc_wino4_conv_impl(d, in, U, bias, out, bnpart, s);
    WinoK k;
    const int rc = fill(d, k);
    if (rc != PC_OK) return rc;
    PC_CHECK_ARG(in && U && out, "pc_wino_conv: null pointer");
    PC_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)U % 16 == 0), "pc_wino_conv: in / U must be 16-byte aligned");
    PC_CHECK_ARG(!(d->flags & PC_F_BIAS) || bias, "pc_wino_conv: bias flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_BNPART) || bnpart, "pc_wino_conv: bnpart flag without pointer");
    PC_CHECK_ARG(!(d->flags & ~(PC_F_BIAS | PC_F_ACCUM | PC_F_BNPART | PC_F_STRIPS)), "pc_wino_conv: unsupported flag");
    PC_CHECK_ARG(d->act == PC_ACT_NONE || d->act == PC_ACT_RELU, "pc_wino_conv: activation");
    k.in = in; k.U = U; k.bias = bias; k.out = out; k.bnpart = bnpart;
#ifdef PICONS_DIAG
    // diagnostic build only (make diag; tools/probe_wino.py): ablations and in-kernel stamps, bits 1-8 give wrong results by design
    static const int var = getenv("PICONS_WINO_VARIANT") ? atoi(getenv("PICONS_WINO_VARIANT")) : 0;
    PC_CHECK_ARG(var != 32 || bnpart, "pc_wino_conv: variant 32 writes its stamps through bnpart");
#endif
    const size_t lds = (size_t)(4 * PLANE + 2 * RPLANE) * sizeof(float);
    const dim3 grid((unsigned)(wino_spatial_blocks(k) * k.nct));
#define WINO_LAUNCH_(V, S)                                                                                                        \
    {                                                                                                                             \
        PC_SET_LDS_ONCE((wino_conv_kernel<V, S>), lds, "wino_conv_kernel");                                                       \
        if (pc_tl_ev_start) hipExtLaunchKernelGGL((wino_conv_kernel<V, S>), grid, dim3(256), lds, (hipStream_t)s, pc_tl_ev_start, pc_tl_ev_stop, 0, k); \
        else hipLaunchKernelGGL((wino_conv_kernel<V, S>), grid, dim3(256), lds, (hipStream_t)s, k);                               \
    }
#define WINO_LAUNCH(V) { if (k.strip) WINO_LAUNCH_(V, true) else WINO_LAUNCH_(V, false) }
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
        default: WINO_LAUNCH(0) break;
    }
#else
    WINO_LAUNCH(0)
#endif
#undef WINO_LAUNCH
#undef WINO_LAUNCH_
    PC_CHECK_LAUNCH("wino_conv_kernel");
    return PC_OK;
}

