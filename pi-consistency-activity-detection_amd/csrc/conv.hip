// Gather-GEMM convolution family for gfx950: fp32 MFMA (v_mfma_f32_32x32x2_f32), LDS-tiled,
// NDHWC activations, [Co][taps][Ci] weights.  One kernel form covers Conv3d/Conv2d forward
// (TF-SAME padding folded into the gather), conv dgrad, ConvTranspose forward as sub-pixel
// parity classes and ConvTranspose dgrad; a second form computes weight gradients.
// Replaces the ATen/cuDNN convolution call sites listed in SURVEY.md §2a (K1, K6, K9-K12).
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <vector>

namespace {

template <int BM, int BN, int WM, int WN, bool FAST, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvK p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;           // [2][BN][LDK]
    int* rinfo = (int*)(Bs + 2 * BN * LDK);    // [BM][4] n,t0,h0,w0
    int* rout = rinfo + BM * 4;                // [BM] output position index or -1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % p.ntiles, mtl = bid / p.ntiles;
    const int g = mtl / p.mtiles_g, lt = mtl % p.mtiles_g;
    const int n0 = nt * BN;
    const float* wbase = p.w + (size_t)g * p.wgstride;          // per-group weights (0 stride = shared)
    const float* bbase = p.bias + (size_t)g * p.bgstride;

    for (int r = tid; r < BM; r += 256) {
        const int lm = lt * BM + r;
        int4 info = make_int4(-1, 0, 0, 0);
        int op = -1;
        if (lm < p.Mg) {
            int m = g * p.Mg + lm;
            int n = 0;
            if (p.flags & PC_F_NFAST) { const int ng = p.N / p.groups; n = g * ng + lm % ng; m = lm / ng; }   // sample fastest within the group
            const int wq = m % p.Wq; m /= p.Wq;
            const int hq = m % p.Hq; m /= p.Hq;
            const int tq = m % p.Tq;
            if (!(p.flags & PC_F_NFAST)) n = m / p.Tq;
            info = make_int4(n, tq * p.istr[0] + p.ioff0[0], hq * p.istr[1] + p.ioff0[1],
                             wq * p.istr[2] + p.ioff0[2]);
            op = ((n * p.To + tq * p.ostr[0] + p.ooff[0]) * p.Ho + hq * p.ostr[1] + p.ooff[1]) * p.Wo +
                 wq * p.ostr[2] + p.ooff[2];
        }
        ((int4*)rinfo)[r] = info;
        rout[r] = op;
    }
    __syncthreads();

    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    int4 ri[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) ri[j] = ((int4*)rinfo)[lrow + 32 * j];

    f32x4 areg[AR], breg[BR];
    const int tapHW = p.ntap[1] * p.ntap[2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Branch-free tile fetch: invalid rows / taps read a safe address and are zeroed by a select, so the
    // compiler emits straight-line global_load_dwordx4 instead of one exec-masked branch per load.
    unsigned vmask = 0;   // validity bits of the in-flight tile (A rows then B rows); applied when the tile is stored to LDS
    auto fetch = [&](int dt, int dh, int dw, int wtap, int ci, bool kval) {
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int t = ri[j].y + dt, h = ri[j].z + dh, w = ri[j].w + dw;
            const bool v = kval && ri[j].x >= 0 && (unsigned)t < (unsigned)p.Ti &&
                           (unsigned)h < (unsigned)p.Hi && (unsigned)w < (unsigned)p.Wi;
            const size_t pos = (size_t)(((ri[j].x * p.Ti + t) * p.Hi + h) * p.Wi + w);
            const float* src = v ? p.in + pos * p.ldi + ci : p.in;
            areg[j] = *(const f32x4*)src;
            m |= (v ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int co = n0 + lrow + 32 * j;
            const bool v = kval && co < p.Co;
            const float* src = v ? wbase + ((size_t)co * p.wtaps + wtap) * p.ldw + ci : wbase;
            breg[j] = *(const f32x4*)src;
            m |= (v ? 1u : 0u) << (AR + j);
        }
        vmask = m;
    };
    // FAST (Ci % 32 == 0): a K chunk lies inside one tap, so the tap walk is wave-uniform scalar state
    // advanced once per chunk; otherwise decode (tap, ci) per thread with integer division.
    int u_a = 0, u_b = 0, u_c = 0, u_ci = 0;
    auto gload = [&](int c) {
        if (FAST) {
            const int dt = u_a * p.istep[0], dh = u_b * p.istep[1], dw = u_c * p.istep[2];
            const int wtap = ((p.wk0[0] + u_a * p.wkstep[0]) * p.KH + p.wk0[1] + u_b * p.wkstep[1]) * p.KW +
                             p.wk0[2] + u_c * p.wkstep[2];
            fetch(dt, dh, dw, wtap, u_ci + lcol, true);
            u_ci += BK;
            if (u_ci >= p.Ci) {
                u_ci = 0;
                if (++u_c == p.ntap[2]) { u_c = 0; if (++u_b == p.ntap[1]) { u_b = 0; ++u_a; } }
            }
        } else {
            const int kk = c * BK + lcol;
            const bool kval = kk < p.K;
            const int tap = kk / p.Ci, ci = kk - tap * p.Ci;
            const int a_ = tap / tapHW, rem = tap - a_ * tapHW;
            const int b_ = rem / p.ntap[2], c_ = rem - b_ * p.ntap[2];
            const int wtap = ((p.wk0[0] + a_ * p.wkstep[0]) * p.KH + p.wk0[1] + b_ * p.wkstep[1]) * p.KW +
                             p.wk0[2] + c_ * p.wkstep[2];
            fetch(a_ * p.istep[0], b_ * p.istep[1], c_ * p.istep[2], kval ? wtap : 0, kval ? ci : 0, kval);
        }
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < AR; ++j) *(f32x4*)(a + (lrow + 32 * j) * LDK + lcol) = ((vmask >> j) & 1u) ? areg[j] : zero4;
#pragma unroll
        for (int j = 0; j < BR; ++j) *(f32x4*)(b + (lrow + 32 * j) * LDK + lcol) = ((vmask >> (AR + j)) & 1u) ? breg[j] : zero4;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int arow = wm * (BM / WM) + (lane & 31), brow = wn * (BN / WN) + (lane & 31);
    const int kof = (lane >> 5) * 4;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = (ABL >= 1) ? 0 : (c & 1);
        if (ABL == 0 && c + 1 < nchunks) gload(c + 1);
        const float* a = As + buf * BM * LDK + arow * LDK + kof;
        const float* b = Bs + buf * BN * LDK + brow * LDK + kof;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(a + i * 32 * LDK + ((ABL >= 3) ? 0 : ks * 8));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4*)(b + j * 32 * LDK + ((ABL >= 3) ? 0 : ks * 8));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (ABL == 0 && c + 1 < nchunks) lstore(buf ^ 1);
        if (ABL <= 1) __syncthreads();
    }

    // ---- epilogue: C/D map row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const bool has_bias = p.flags & PC_F_BIAS, has_cs = p.flags & PC_F_CSCALE, accum = p.flags & PC_F_ACCUM;
    if (p.flags & PC_F_BNPART) {
        float* part = p.bnpart + ((size_t)(g * p.mtiles_g + lt) * WM + wm) * 2 * p.Co;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; s2 += v * v; }
            s += __shfl_xor(s, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (lane < 32 && col < p.Co) { part[col] = s; part[p.Co + col] = s2; }
        }
    }
    if (ABL == 0 && (p.flags & PC_F_TOUT)) { store_tile_cols<BM, BN, WM, WN, TM, TN>(acc, smem, rout, rinfo, p, n0, wm, wn, lane, tid); return; }
    if (ABL == 0 && store_tile_rows<BM, BN, WM, WN, TM, TN>(acc, smem, rout, rinfo, p, bbase, n0, wm, wn, lane, tid)) return;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int op = rout[row];
            if (op < 0) continue;
            const int nb = rinfo[row * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
                if (col >= p.Co) continue;
                float v = acc[i][j][r];
                if (has_bias) v += bbase[col];
                if (col >= p.act_c0) {
                    if (p.act == PC_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == PC_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                }
                if (has_cs) v *= p.cscale[(size_t)nb * p.Co + col];
                float* o = p.out + (size_t)op * p.ldo + col;
                if (accum) v += *o;
                *o = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path (Ci % 32 == 0, <= 10 taps per dim): tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR
// staging, no ds_write), per-row pointers and per-dim tap-validity bit masks are computed once per block, and the
// tap walk is wave-uniform scalar state, so a K chunk costs ~6 VALU per 1 KiB fetched instead of ~20.  The LDS image
// is un-padded [row][32 floats] with the 16-byte slots of a row XOR-swizzled by (row>>1)&7; the swizzle is applied on
// the SOURCE address (LDS-DMA writes lane-linear) and again on the fragment read, which keeps ds_read_b128
// conflict-free for both 16-lane groups.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <int BM, int BN, int WM, int WN, int VAR = 0>
__global__ __launch_bounds__(256, 2) void conv_gemm_glds_kernel(const ConvK p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    // VAR bits 3-4: LDS-DMA ring of 2 + n stages instead of the double buffer.  With one chunk in flight a block waits out the
    // full latency of every tile fetch that misses its XCD's L2 (the activations were just written by a kernel on other XCDs)
    // unless a second resident block covers it -- which the under-filled 28x28 launches (392 blocks on 512 slots) do not have.
    constexpr int ST = 2 + ((VAR >> 3) & 3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [ST][BM][32]
    float* Bs = smem + ST * BM * BK;           // [ST][BN][32]
    int* rinfo = (int*)(Bs + ST * BN * BK);    // [BM][4] n,t0,h0,w0
    int* rout = rinfo + BM * 4;                // [BM] output position index or -1
    unsigned* tile_or = (unsigned*)(rout + BM);   // OR of every row's tap-validity mask

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // VAR bit 2 (TAP4): Ci == 4 (the stem: 3 channels padded to one 16-byte piece per tap) - a K chunk is 8 consecutive taps of the
    // tile's tap box instead of 32 channels of one tap, so every 16-byte DMA piece has its own tap (address, validity)
    constexpr bool TAP4 = (VAR & 4) != 0;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % p.ntiles, mtl = bid / p.ntiles;
    const int g = mtl / p.mtiles_g, lt = mtl % p.mtiles_g;
    const int n0 = nt * BN;
    const float* wbase = p.w + (size_t)g * p.wgstride;
    const float* bbase = p.bias + (size_t)g * p.bgstride;
    if (tid == 0) *tile_or = 0u;

    for (int r = tid; r < BM; r += 256) {
        const int lm = lt * BM + r;
        int4 info = make_int4(-1, 0, 0, 0);
        int op = -1;
        if (lm < p.Mg) {
            int m = g * p.Mg + lm;
            int n = 0;
            if (p.flags & PC_F_NFAST) { const int ng = p.N / p.groups; n = g * ng + lm % ng; m = lm / ng; }   // sample fastest within the group
            const int wq = m % p.Wq; m /= p.Wq;
            const int hq = m % p.Hq; m /= p.Hq;
            const int tq = m % p.Tq;
            if (!(p.flags & PC_F_NFAST)) n = m / p.Tq;
            info = make_int4(n, tq * p.istr[0] + p.ioff0[0], hq * p.istr[1] + p.ioff0[1], wq * p.istr[2] + p.ioff0[2]);
            op = ((n * p.To + tq * p.ostr[0] + p.ooff[0]) * p.Ho + hq * p.ostr[1] + p.ooff[1]) * p.Wo + wq * p.ostr[2] + p.ooff[2];
        }
        ((int4*)rinfo)[r] = info;
        rout[r] = op;
    }
    __syncthreads();

    // per-thread fetch rows: row = lrow + 32*j, 16-byte slot (tid&7), source slot XOR-swizzled
    const int lrow = tid >> 3, slot = tid & 7;
    unsigned adma[AR], amask[AR];      // byte offset of the row's tap-origin position from (p.in - abias): never negative
    unsigned bdma[BR];                 // byte offset of the output channel's weight row from wbase, DMA_OOB past Co
    // tap origins may lie in the padding (negative coordinates): the lowest origin position any row can have, as a bias on the base
    const long long apos0 = ((long long)min(p.ioff0[0], 0) * p.Hi + min(p.ioff0[1], 0)) * p.Wi + min(p.ioff0[2], 0);
    const long long abias = -apos0 * p.ldi;      // floats
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = lrow + 32 * j;
        const int4 ri = ((int4*)rinfo)[row];
        const int ks = TAP4 ? 0 : (slot ^ ((row >> 1) & 7)) * 4;
        unsigned m = 0;
        if (ri.x >= 0) {
            for (int a = 0; a < p.ntap[0]; ++a) m |= ((unsigned)(ri.y + a * p.istep[0]) < (unsigned)p.Ti ? 1u : 0u) << a;
            for (int a = 0; a < p.ntap[1]; ++a) m |= ((unsigned)(ri.z + a * p.istep[1]) < (unsigned)p.Hi ? 1u : 0u) << (10 + a);
            for (int a = 0; a < p.ntap[2]; ++a) m |= ((unsigned)(ri.w + a * p.istep[2]) < (unsigned)p.Wi ? 1u : 0u) << (20 + a);
        }
        amask[j] = m;
        const long long pos = ((long long)(ri.x * p.Ti + ri.y) * p.Hi + ri.z) * p.Wi + ri.w;   // may be "out of range": only used when valid
        adma[j] = (unsigned)(((pos - apos0) * p.ldi + ks) * 4);
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int row = lrow + 32 * j, co = n0 + row;
        const int ks = TAP4 ? 0 : (slot ^ ((row >> 1) & 7)) * 4;
        bdma[j] = co < p.Co ? (unsigned)(((size_t)co * p.wtaps * p.ldw + ks) * 4) : DMA_OOB;
    }

    // Tap box of the tile: per dimension a row's valid taps form an interval, and so does their union over the tile.
    // Taps outside the box gather only padding for EVERY row, so the K loop walks the box instead of all taps.
    {
        unsigned mo = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) mo |= amask[j];
        atomicOr(tile_or, mo);
    }
    __syncthreads();
    const unsigned bo = __builtin_amdgcn_readfirstlane(*tile_or);
    const unsigned bt = bo & 0x3ffu, bh = (bo >> 10) & 0x3ffu, bw = (bo >> 20) & 0x3ffu;
    const bool any_tap = bt && bh && bw;
    const int a_lo = any_tap ? __builtin_ctz(bt) : 0, a_hi = any_tap ? 31 - __builtin_clz(bt) : -1;
    const int b_lo = any_tap ? __builtin_ctz(bh) : 0, b_hi = any_tap ? 31 - __builtin_clz(bh) : -1;
    const int c_lo = any_tap ? __builtin_ctz(bw) : 0, c_hi = any_tap ? 31 - __builtin_clz(bw) : -1;
    int u_a = a_lo, u_b = b_lo, u_c = c_lo, u_ci = 0;
    // TAP4: this thread's logical 16-byte slot of a chunk (the same for all its rows: the swizzle depends on lrow only)
    // is tap (8 * chunk + ksl) of the box in (a, b, c) order
    const int nb_ = b_hi - b_lo + 1, nc_ = c_hi - c_lo + 1;
    int t_a = a_lo, t_b = b_lo, t_c = c_lo;
    if (TAP4 && any_tap) {
        const int ksl = slot ^ ((lrow >> 1) & 7);
        t_c = c_lo + ksl % nc_;
        const int r = ksl / nc_;
        t_b = b_lo + r % nb_; t_a = a_lo + r / nb_;
    }
    auto fetch = [&](int buf) {
        if (TAP4) {
            const bool inb = t_a <= a_hi;
            const long long da = ((long long)(t_a * p.istep[0] * p.Hi + t_b * p.istep[1]) * p.Wi + t_c * p.istep[2]) * p.ldi;
            const unsigned sel = inb ? ((1u << t_a) | (1u << (10 + t_b)) | (1u << (20 + t_c))) : 0xffffffffu;   // never matches out of the box
            const int wtap = ((p.wk0[0] + t_a * p.wkstep[0]) * p.KH + p.wk0[1] + t_b * p.wkstep[1]) * p.KW + p.wk0[2] + t_c * p.wkstep[2];
            const long long db = (long long)wtap * p.ldw;
            float* la = As + buf * BM * BK + (wave * 8) * BK;
            float* lb = Bs + buf * BN * BK + (wave * 8) * BK;
            // every lane has its own tap here: the tap offset is a 32-bit per-lane add (wraps back into range whenever the tap is valid)
            const dma_rsrc_t ra = dma_rsrc(p.in - abias), rb = dma_rsrc(wbase);
            const unsigned dab = (unsigned)(da * 4), dbb = (unsigned)(db * 4);
#pragma unroll
            for (int j = 0; j < AR; ++j) glds16b(ra, ((amask[j] & sel) == sel) ? adma[j] + dab : DMA_OOB, la + j * 32 * BK);
#pragma unroll
            for (int j = 0; j < BR; ++j) glds16b(rb, (bdma[j] != DMA_OOB && inb) ? bdma[j] + dbb : DMA_OOB, lb + j * 32 * BK);
            t_c += 8;
            while (t_c > c_hi) { t_c -= nc_; if (++t_b > b_hi) { t_b = b_lo; ++t_a; } }
            return;
        }
        const long long da = ((long long)(u_a * p.istep[0] * p.Hi + u_b * p.istep[1]) * p.Wi + u_c * p.istep[2]) * p.ldi + u_ci;
        const unsigned sel = (1u << u_a) | (1u << (10 + u_b)) | (1u << (20 + u_c));
        const int wtap = ((p.wk0[0] + u_a * p.wkstep[0]) * p.KH + p.wk0[1] + u_b * p.wkstep[1]) * p.KW + p.wk0[2] + u_c * p.wkstep[2];
        const long long db = (long long)wtap * p.ldw + u_ci;
        float* la = As + buf * BM * BK + (wave * 8) * BK;     // wave-uniform base; the DMA adds lane*16 B itself
        float* lb = Bs + buf * BN * BK + (wave * 8) * BK;
        // the tap / channel-chunk offset is wave-uniform: it moves the resource base (scalar adds), the lane offsets never change
        const dma_rsrc_t ra = dma_rsrc(p.in + (da - abias)), rb = dma_rsrc(wbase + db);
#pragma unroll
        for (int j = 0; j < AR; ++j) glds16b(ra, ((amask[j] & sel) == sel) ? adma[j] : DMA_OOB, la + j * 32 * BK);
#pragma unroll
        for (int j = 0; j < BR; ++j) glds16b(rb, bdma[j], lb + j * 32 * BK);
        u_ci += BK;
        if (u_ci >= p.Ci) {
            u_ci = 0;
            if (++u_c > c_hi) { u_c = c_lo; if (++u_b > b_hi) { u_b = b_lo; ++u_a; } }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = TAP4 ? ((a_hi - a_lo + 1) * nb_ * nc_ + 7) / 8 : (a_hi - a_lo + 1) * nb_ * nc_ * (p.Ci / BK);
    const int arow = wm * (BM / WM) + (lane & 31), brow = wn * (BN / WN) + (lane & 31);
    const int kh = lane >> 5;
    int aoff[TM], boff[TN], asw[TM], bsw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = arow + i * 32; aoff[i] = r * BK; asw[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = brow + j * 32; boff[j] = r * BK; bsw[j] = (r >> 1) & 7; }
    auto mma_chunk = [&](int buf, int c) {
        const float* a = As + buf * BM * BK;
        const float* b = Bs + buf * BN * BK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 af[TM], bf[TN];
            const int q = ks * 2 + kh;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(a + aoff[i] + ((q ^ asw[i]) << 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const f32x4*)(b + boff[j] + ((q ^ bsw[j]) << 2));
            if (VAR & 1) __builtin_amdgcn_s_setprio(1);
            // TAP4 with PC_F_CI3 (VAR bit 5): a 16-byte slot is one tap's (c0, c1, c2, padding) -- the padding channel's MFMA is not issued
            constexpr int NE = (TAP4 && (VAR & 32)) ? 3 : 4;
#pragma unroll
            for (int e = 0; e < NE; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
            if (ST == 2 && (VAR & 2) && ks == 0 && c + 1 < nchunks) fetch(buf ^ 1);   // issue the next tile's DMA behind the first MFMA group
        }
    };
    if constexpr (ST == 2) {
        if (nchunks > 0) fetch(0);
        PC_SYNC_DMA();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (!(VAR & 2) && c + 1 < nchunks) fetch(buf ^ 1);
            mma_chunk(buf, c);
            PC_SYNC_DMA();        // the LDS-DMA of chunk c+1 has landed (vmcnt(0), written out: common.h) and the reads of chunk c are fenced
        }
    } else {
        // ring: chunks c .. c+ST-2 in flight while chunk c is multiplied.  Each thread issues AR + BR DMA pieces per chunk, in order,
        // so "chunk c has landed" is vmcnt <= (chunks issued after c) * (AR + BR) for every wave, then the barrier (which also says
        // every wave is done reading the buffer the next fetch overwrites: the one multiplied in the previous iteration).
        constexpr int PIECES = AR + BR;
        for (int c = 0; c < ST - 1 && c < nchunks; ++c) fetch(c);
        int buf = 0, fbuf = (ST - 1) % ST;
        for (int c = 0; c < nchunks; ++c) {
            if (nchunks - 1 - c >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((ST - 2) * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (c + ST - 1 < nchunks) fetch(fbuf);
            mma_chunk(buf, c);
            buf = buf + 1 == ST ? 0 : buf + 1;
            fbuf = fbuf + 1 == ST ? 0 : fbuf + 1;
        }
        __syncthreads();          // the operand ring is reused as the output staging tile
    }

    // ---- epilogue (same as conv_gemm_kernel)
    const bool has_bias = p.flags & PC_F_BIAS, has_cs = p.flags & PC_F_CSCALE, accum = p.flags & PC_F_ACCUM;
    if (p.flags & PC_F_BNPART) {
        float* part = p.bnpart + ((size_t)(g * p.mtiles_g + lt) * WM + wm) * 2 * p.Co;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; s2 += v * v; }
            s += __shfl_xor(s, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (lane < 32 && col < p.Co) { part[col] = s; part[p.Co + col] = s2; }
        }
    }
    static_assert(BM * BN <= 2 * (BM + BN) * BK, "output tile fits the operand buffers");
    if (p.flags & PC_F_TOUT) { store_tile_cols<BM, BN, WM, WN, TM, TN>(acc, smem, rout, rinfo, p, n0, wm, wn, lane, tid); return; }
    if (store_tile_rows<BM, BN, WM, WN, TM, TN>(acc, smem, rout, rinfo, p, bbase, n0, wm, wn, lane, tid)) return;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int op = rout[row];
            if (op < 0) continue;
            const int nb = rinfo[row * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
                if (col >= p.Co) continue;
                float v = acc[i][j][r];
                if (has_bias) v += bbase[col];
                if (col >= p.act_c0) {
                    if (p.act == PC_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == PC_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                }
                if (has_cs) v *= p.cscale[(size_t)nb * p.Co + col];
                float* o = p.out + (size_t)op * p.ldo + col;
                if (accum) v += *o;
                *o = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int VAR>
int launch_conv_glds_v(const ConvK& k, hipStream_t s) {
    constexpr int ST = 2 + ((VAR >> 3) & 3);
    {   // the tiles are fetched with 32-bit lane offsets from a wave-uniform base (glds16b)
        const long long halo = ((long long)(k.ioff0[0] < 0 ? -k.ioff0[0] : 0) * k.Hi + (k.ioff0[1] < 0 ? -k.ioff0[1] : 0)) * k.Wi + (k.ioff0[2] < 0 ? -k.ioff0[2] : 0);
        const long long in_bytes = ((long long)k.N * k.Ti * k.Hi * k.Wi + 2 * halo) * k.ldi * 4, w_bytes = (long long)k.Co * k.wtaps * k.ldw * 4;
        PC_CHECK_ARG(in_bytes < DMA_MAX_BYTES && w_bytes < DMA_MAX_BYTES,
                     "pc_conv_fwd: input (%lld B) or weights (%lld B) exceed the 4 GiB the LDS-DMA gather addresses: use a smaller per-GPU batch", in_bytes, w_bytes);
    }
    const size_t lds = (size_t)(ST * (BM + BN) * BK + BM * 5 + 4) * sizeof(float);
    PC_SET_LDS_ONCE((conv_gemm_glds_kernel<BM, BN, WM, WN, VAR>), lds, "conv_gemm_glds_kernel");
    ConvK p = k;
    p.mtiles_g = cdiv(p.Mg, BM);
    p.ntiles = cdiv(p.Co, BN);
    if (pc_tl_ev_start)
        hipExtLaunchKernelGGL((conv_gemm_glds_kernel<BM, BN, WM, WN, VAR>), dim3(p.groups * p.mtiles_g * p.ntiles), dim3(256), lds, s, pc_tl_ev_start, pc_tl_ev_stop, 0, p);
    else
        hipLaunchKernelGGL((conv_gemm_glds_kernel<BM, BN, WM, WN, VAR>), dim3(p.groups * p.mtiles_g * p.ntiles), dim3(256), lds, s, p);
    PC_CHECK_LAUNCH("conv_gemm_glds_kernel");
    return PC_OK;
}

// PICONS_CONV_VARIANT (tuning, tools/ablate_conv.py): bit 0 = s_setprio around the MFMA groups, bit 1 = issue the next
// tile's LDS-DMA after the first MFMA group instead of before it.  Same results in every variant.
template <int BM, int BN, int WM, int WN>
int launch_conv_glds(const ConvK& k, hipStream_t s) {
    static const int var = getenv("PICONS_CONV_VARIANT") ? atoi(getenv("PICONS_CONV_VARIANT")) : CONV_DEFAULT_VARIANT;
    // LDS-DMA ring depth.  Default: 3 stages for launches of at most one block per CU (the 196-block 28x28 layers: 10-25 % faster --
    // nothing else on the CU covers the latency of a tile fetch that misses L2), the double buffer otherwise (a ring costs full
    // launches 5-8 %).  PICONS_CONV_STAGES=2|3|4 forces a depth where two blocks of it still fit a CU.
    static const int stages = getenv("PICONS_CONV_STAGES") ? atoi(getenv("PICONS_CONV_STAGES")) : 0;
    if constexpr (BM * BN <= 128 * 64) {
        const size_t per_stage = (size_t)(BM + BN) * BK * sizeof(float);
        const long long grid = (long long)k.groups * cdiv(k.Mg, BM) * cdiv(k.Co, BN);
        const int want = stages ? stages : (grid <= 256 ? 3 : 2);
        if (want == 4 && 4 * per_stage <= 72 * 1024) return launch_conv_glds_v<BM, BN, WM, WN, 16>(k, s);
        if (want >= 3 && 3 * per_stage <= 76 * 1024) return launch_conv_glds_v<BM, BN, WM, WN, 8>(k, s);
    }
    switch (var & 3) {
        case 1: return launch_conv_glds_v<BM, BN, WM, WN, 1>(k, s);
        case 2: return launch_conv_glds_v<BM, BN, WM, WN, 2>(k, s);
        case 3: return launch_conv_glds_v<BM, BN, WM, WN, 3>(k, s);
        default: return launch_conv_glds_v<BM, BN, WM, WN, 0>(k, s);
    }
}

template <int BM, int BN, int WM, int WN, bool FAST>
int launch_conv2(const ConvK& k, hipStream_t s) {
    const size_t lds = (size_t)(2 * (BM + BN) * LDK + BM * 5) * sizeof(float);
    PC_SET_LDS_ONCE((conv_gemm_kernel<BM, BN, WM, WN, FAST>), lds, "conv_gemm_kernel");
    ConvK p = k;
    p.mtiles_g = cdiv(p.Mg, BM);
    p.ntiles = cdiv(p.Co, BN);
    const int grid = p.groups * p.mtiles_g * p.ntiles;
    if (pc_tl_ev_start) hipExtLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, FAST>), dim3(grid), dim3(256), lds, s, pc_tl_ev_start, pc_tl_ev_stop, 0, p);
    else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, FAST>), dim3(grid), dim3(256), lds, s, p);
    PC_CHECK_LAUNCH("conv_gemm_kernel");
    return PC_OK;
}

#ifdef PICONS_DIAG
// Diagnostic build only (make diag -> libpicons_diag.so; the product library neither holds these kernels nor reads the variable).
// PICONS_CONV_ABLATE=1|2|3 (tools/ablate_conv.py): 1 = no tile fetch / LDS store in the K loop,
// 2 = also no barrier, 3 = also a fixed fragment address.  Results are WRONG in these modes; they only price phases.
template <int BM, int BN, int WM, int WN, int ABL>
int launch_ablate(const ConvK& k, hipStream_t s) {
    const size_t lds = (size_t)(2 * (BM + BN) * LDK + BM * 5) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<BM, BN, WM, WN, true, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ConvK p = k;
    p.mtiles_g = cdiv(p.Mg, BM);
    p.ntiles = cdiv(p.Co, BN);
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, true, ABL>), dim3(p.groups * p.mtiles_g * p.ntiles), dim3(256), lds, s, p);
    PC_CHECK_LAUNCH("conv_gemm_kernel(ablate)");
    return PC_OK;
}
#endif

template <int BM, int BN, int WM, int WN>
int launch_conv(const ConvK& k, hipStream_t s) {
#ifdef PICONS_DIAG
    static const int abl = getenv("PICONS_CONV_ABLATE") ? atoi(getenv("PICONS_CONV_ABLATE")) : 0;
    if (abl && k.Ci % BK == 0 && BM == 128 && BN >= 64) {
        if (abl == 1) return launch_ablate<BM, BN, WM, WN, 1>(k, s);
        if (abl == 2) return launch_ablate<BM, BN, WM, WN, 2>(k, s);
        return launch_ablate<BM, BN, WM, WN, 3>(k, s);
    }
#endif
    static const int no_glds = getenv("PICONS_CONV_NO_GLDS") ? atoi(getenv("PICONS_CONV_NO_GLDS")) : 0;
    const bool fast = k.Ci % BK == 0;
    static const int no_tap4 = getenv("PICONS_CONV_NO_TAP4") ? atoi(getenv("PICONS_CONV_NO_TAP4")) : 0;
    if constexpr (BM == 128 && BN == 64)
        if (k.Ci == 4 && !no_glds && !no_tap4 && k.ntap[0] <= 10 && k.ntap[1] <= 10 && k.ntap[2] <= 10 && ((uintptr_t)k.in % 16 == 0) && k.ldi % 4 == 0 &&
            k.ldw % 4 == 0)
            return (k.flags & PC_F_CI3) ? launch_conv_glds_v<BM, BN, WM, WN, 4 | 32>(k, s) : launch_conv_glds_v<BM, BN, WM, WN, 4>(k, s);
    if (fast && !no_glds && k.ntap[0] <= 10 && k.ntap[1] <= 10 && k.ntap[2] <= 10 && ((uintptr_t)k.in % 16 == 0) && k.ldi % 4 == 0)
        return launch_conv_glds<BM, BN, WM, WN>(k, s);
    return fast ? launch_conv2<BM, BN, WM, WN, true>(k, s) : launch_conv2<BM, BN, WM, WN, false>(k, s);
}

// tile choice: minimise padded work, prefer larger tiles when the grid still fills the chip
struct TileCfg { int bm, bn, wm; };
inline TileCfg choose_tile(int Mg, int groups, int Co, int Ci) {
    if (Ci == 4) return {128, 64, 2};          // one 16-byte piece per tap: the 8-taps-per-chunk LDS-DMA variant has this tile only
    const TileCfg cand[] = {{128, 128, 2}, {128, 64, 2}, {64, 64, 2}, {128, 32, 4}};
    double best = 1e30; TileCfg bc = cand[0];
    for (const TileCfg& c : cand) {
        const double mt = (double)groups * cdiv(Mg, c.bm), ntl = cdiv(Co, c.bn);
        const double blocks = mt * ntl;
        const double work = blocks * c.bm * c.bn;                 // padded MACs per unit K
        const double eff = (c.bm * c.bn) / (double)(c.bm + c.bn); // operand reuse
        // waves of blocks over 512 slots (2 blocks/CU); partial last wave costs a full one
        const double slots = 512.0, wavesq = ceil(blocks / slots);
        const double fill = blocks / (wavesq * slots);
        const double cost = work / (fill > 0.9 ? 1.0 : fill / 0.9) * (1.0 + 8.0 / eff);   // swept 0.35..1.0 on the step
        if (cost < best) { best = cost; bc = c; }
    }
    return bc;
}

// The tile pc_conv_fwd launches for a descriptor (shared with the host-side work accounting, pc_conv_work).
inline TileCfg launch_tile(const pc_conv_desc* d, int groups) {
    const int64_t Mg = (int64_t)(d->N / groups) * d->Tq * d->Hq * d->Wq;
    TileCfg c = choose_tile((int)Mg, groups, d->Co, d->Ci);
    static const char* force = getenv("PICONS_CONV_TILE");      // diagnostic: "bm,bn" for grouped launches without BN partials
    if (force && groups > 2 && !(d->flags & PC_F_BNPART)) {
        int bm = 0, bn = 0;
        if (sscanf(force, "%d,%d", &bm, &bn) == 2) { c.bm = bm; c.bn = bn; c.wm = (bm == 64 && bn == 128) ? 1 : (bn == 32 ? 4 : 2); }
    }
    if (d->flags & PC_F_NFAST) {
        // n-fastest rows: a tile should hold whole groups of N samples of consecutive w so its tap box is tight;
        // when Wq*N is not a multiple of 128 use 64-row tiles (28 w x 16 samples = 7 tiles of 64, none straddles a row)
        const long long per_row = (long long)d->Wq * (d->N / groups);
        if (per_row % 128 != 0 && per_row % 64 == 0) { c.bm = 64; c.bn = d->Co >= 128 ? 128 : 64; c.wm = d->Co >= 128 ? 1 : 2; }
    }
    return c;
}

}  // namespace

// Host-only accounting of the multiply-accumulates one pc_conv_fwd launch performs (no GPU call).  It walks the launch's tiles
// with the kernel's own row order (conv_gemm_glds_kernel: rinfo / amask / tap box) and counts, per tile, the K the block's loop
// really walks.  out[0] = MACs ISSUED to the matrix cores (whole BM x BN tiles, what an MFMA instruction counter sees),
// out[1] = MACs EXECUTED on real outputs (real rows x real columns x the tile's K: padding taps INSIDE the tap box included),
// out[2] = VALID MACs (taps that read inside the volume only, real channels), out[3] = blocks, out[4] = BM, out[5] = BN,
// out[6] = 1 if the LDS-DMA kernel (tap box) runs, 0 for the register-staged kernel (flattened K, no tap skipping).
// ci_real / co_real: channels that are not padding (0 = Ci / Co).
extern "C" int pc_conv_work(const pc_conv_desc* d, int ci_real, int co_real, double* out) {
    PC_CHECK_ARG(d && out, "pc_conv_work: null pointer");
    const int groups = d->groups > 0 ? d->groups : 1;
    PC_CHECK_ARG(d->N % groups == 0, "pc_conv_work: N %% groups");
    const int64_t Mg = (int64_t)(d->N / groups) * d->Tq * d->Hq * d->Wq;
    TileCfg c = launch_tile(d, groups);
    const bool x6 = pc_x6_eligible(d);                   // PC_F_X6: the bf16-split kernel's tiles (conv_x6.hip), always with the tap box
    if (x6) { const X6Tile t = pc_x6_tile(d, groups); c.bm = t.bm; c.bn = t.bn; c.wm = t.wm; }
    if (ci_real <= 0) ci_real = d->Ci;
    if (co_real <= 0) co_real = d->Co;
    const bool small_taps = d->ntap[0] <= 10 && d->ntap[1] <= 10 && d->ntap[2] <= 10;
    static const int no_glds = getenv("PICONS_CONV_NO_GLDS") ? atoi(getenv("PICONS_CONV_NO_GLDS")) : 0;
    static const int no_tap4 = getenv("PICONS_CONV_NO_TAP4") ? atoi(getenv("PICONS_CONV_NO_TAP4")) : 0;
    const bool tap4 = !x6 && c.bm == 128 && c.bn == 64 && d->Ci == 4 && !no_glds && !no_tap4 && small_taps;
    const bool glds = x6 || tap4 || (d->Ci % BK == 0 && !no_glds && small_taps);
    const int64_t mtiles = cdiv(Mg, c.bm), ntiles = cdiv(d->Co, c.bn);
    const int ng = d->N / groups;
    const int I[3] = {d->Ti, d->Hi, d->Wi};
    double issued = 0, executed = 0, valid = 0;
    const int Kflat = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Ci;
    for (int g = 0; g < groups; ++g)
        for (int64_t lt = 0; lt < mtiles; ++lt) {
            unsigned box[3] = {0, 0, 0};
            int64_t rows = 0;
            double vtaps = 0;
            for (int r = 0; r < c.bm; ++r) {
                const int64_t lm = lt * c.bm + r;
                if (lm >= Mg) break;
                ++rows;
                int64_t m = (int64_t)g * Mg + lm;
                if (d->flags & PC_F_NFAST) m = lm / ng;
                const int q[3] = {(int)((m / ((int64_t)d->Wq * d->Hq)) % d->Tq), (int)((m / d->Wq) % d->Hq), (int)(m % d->Wq)};
                int nv[3];
                for (int k = 0; k < 3; ++k) {
                    nv[k] = 0;
                    const int base = q[k] * d->istr[k] + d->ioff0[k];
                    for (int a = 0; a < d->ntap[k]; ++a)
                        if ((unsigned)(base + a * d->istep[k]) < (unsigned)I[k]) { box[k] |= 1u << a; ++nv[k]; }
                }
                vtaps += (double)nv[0] * nv[1] * nv[2];
            }
            double Ktile, Kreal;                     // K the block walks (kernel channels) / the same in real channels
            if (glds) {
                int ext[3];
                bool any = box[0] && box[1] && box[2];
                for (int k = 0; k < 3; ++k) ext[k] = any ? (32 - __builtin_clz(box[k])) - __builtin_ctz(box[k]) : 0;
                const double taps = (double)ext[0] * ext[1] * ext[2];
                if (tap4) {                          // 8 taps per chunk; with PC_F_CI3 three MFMAs per 16-byte piece instead of four
                    const double ch = (d->flags & PC_F_CI3) ? 3.0 : 4.0;
                    Ktile = ceil(taps / 8.0) * 8.0 * ch;
                    Kreal = taps * ci_real;
                } else {
                    Ktile = taps * d->Ci;
                    Kreal = taps * ci_real;
                }
            } else {
                Ktile = (double)cdiv(Kflat, BK) * BK;
                Kreal = (double)d->ntap[0] * d->ntap[1] * d->ntap[2] * ci_real;
            }
            // the bf16-split kernel: a wave owns 32 rows of the tile and multiplies nothing when none of them is a real row
            issued += (double)(x6 ? cdiv(rows, 32) * 32 : c.bm) * c.bn * ntiles * Ktile;
            executed += (double)rows * co_real * Kreal;
            valid += vtaps * ci_real * co_real;
        }
    out[0] = issued; out[1] = executed; out[2] = valid; out[3] = (double)groups * mtiles * ntiles; out[4] = c.bm; out[5] = c.bn; out[6] = glds ? 1.0 : 0.0;
    return PC_OK;
}

extern "C" int pc_conv_bnpart_rows(const pc_conv_desc* d) {
    const int groups = d->groups > 0 ? d->groups : 1;
    const int64_t Mg = (int64_t)(d->N / groups) * d->Tq * d->Hq * d->Wq;
    if (pc_x6_eligible(d)) { const X6Tile t = pc_x6_tile(d, groups); return groups * cdiv(Mg, t.bm) * t.wm; }
    const TileCfg c = choose_tile((int)Mg, groups, d->Co, d->Ci);
    return groups * cdiv(Mg, c.bm) * c.wm;
}

static int pc_conv_fwd_g(const pc_conv_desc* d, int groups, const float* in, const float* w, const float* bias,
                  const float* cscale, float* out, float* bnpart, hipStream_t s) {
    PC_CHECK_ARG(d && in && w && out, "pc_conv_fwd: null pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_X6), "pc_conv_fwd: PC_F_X6 descriptors go to pc_conv_fwd_x6 (weight planes)");
    PC_CHECK_ARG(d->Ci % 4 == 0 && d->ldi % 4 == 0 && d->ldw % 4 == 0, "pc_conv_fwd: Ci/ldi/ldw must be multiples of 4 (Ci=%d ldi=%d ldw=%d)", d->Ci, d->ldi, d->ldw);
    PC_CHECK_ARG(groups >= 1 && d->N % groups == 0, "pc_conv_fwd: N %% groups");
    PC_CHECK_ARG(!(d->flags & PC_F_BIAS) || bias, "pc_conv_fwd: bias flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_CSCALE) || cscale, "pc_conv_fwd: cscale flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_BNPART) || bnpart, "pc_conv_fwd: bnpart flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_NFAST) || !(d->flags & (PC_F_BNPART | PC_F_CSCALE)), "pc_conv_fwd: NFAST cannot be combined with BN partials / cscale");
    PC_CHECK_ARG(!(d->flags & PC_F_TOUT) || (!(d->flags & (PC_F_BNPART | PC_F_CSCALE | PC_F_BIAS | PC_F_ACCUM | PC_F_NFAST)) && d->act == PC_ACT_NONE),
                 "pc_conv_fwd: channel-major output (PC_F_TOUT) is for plain launches only");
    PC_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)w % 16 == 0), "pc_conv_fwd: in/w must be 16-byte aligned");
    ConvK k;
    k.in = in; k.w = w; k.bias = bias; k.cscale = cscale; k.out = out; k.bnpart = bnpart;
    k.N = d->N; k.Ti = d->Ti; k.Hi = d->Hi; k.Wi = d->Wi; k.Ci = d->Ci; k.ldi = d->ldi;
    k.Tq = d->Tq; k.Hq = d->Hq; k.Wq = d->Wq; k.To = d->To; k.Ho = d->Ho; k.Wo = d->Wo; k.Co = d->Co; k.ldo = d->ldo;
    for (int i = 0; i < 3; ++i) {
        k.ostr[i] = d->ostr[i]; k.ooff[i] = d->ooff[i]; k.istr[i] = d->istr[i]; k.ntap[i] = d->ntap[i];
        k.ioff0[i] = d->ioff0[i]; k.istep[i] = d->istep[i]; k.wk0[i] = d->wk0[i]; k.wkstep[i] = d->wkstep[i];
        PC_CHECK_ARG(d->ntap[i] >= 1, "pc_conv_fwd: ntap < 1");
    }
    k.KH = d->KH; k.KW = d->KW; k.wtaps = d->KT * d->KH * d->KW; k.ldw = d->ldw;
    k.K = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Ci;
    const int64_t M = (int64_t)d->N * d->Tq * d->Hq * d->Wq;
    PC_CHECK_ARG(M > 0 && M < (1ll << 31) && (int64_t)d->N * d->To * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->Ti * d->Hi * d->Wi < (1ll << 31), "pc_conv_fwd: position count out of range");
    k.M = (int)M; k.groups = groups; k.Mg = (int)(M / groups);
    static const int scalar_epi = getenv("PICONS_CONV_SCALAR_EPI") ? atoi(getenv("PICONS_CONV_SCALAR_EPI")) : 0;
    k.act = d->act; k.flags = (d->flags & ~F_SCALAR_EPI) | (scalar_epi ? F_SCALAR_EPI : 0); k.act_c0 = d->act_c0; k.wgstride = d->wgstride; k.bgstride = d->bgstride;
    const TileCfg c = launch_tile(d, groups);
    if (c.bm == 64 && c.bn == 128) return launch_conv<64, 128, 1, 4>(k, s);
    if (c.bm == 128 && c.bn == 128) return launch_conv<128, 128, 2, 2>(k, s);
    if (c.bm == 128 && c.bn == 64) return launch_conv<128, 64, 2, 2>(k, s);
    if (c.bm == 64 && c.bn == 64) return launch_conv<64, 64, 2, 2>(k, s);
    return launch_conv<128, 32, 4, 1>(k, s);
}

extern "C" int pc_conv_fwd(const pc_conv_desc* d, const float* in, const float* w, const float* bias,
                           const float* cscale, float* out, float* bnpart, pc_stream s) {
    return pc_conv_fwd_g(d, d && d->groups > 0 ? d->groups : 1, in, w, bias, cscale, out, bnpart, (hipStream_t)s);
}

// =============================================================================================
// Weight gradient: G[m][n] += sum_pos D[pos][m] * S[gather(pos, tap(n))][cs(n)], n = tap*Cs + cs.
// K (positions) is split over blockIdx.z; fp32 atomics combine the slices.
namespace {

struct WgK {
    const float* D; const float* S; float* g;
    int N, Tq, Hq, Wq, Cd, ldd, Ts, Hs, Ws, Cs, lds;
    int istr[3], ntap[3], ioff0[3], istep[3], wk0[3];
    int KH, KW, NtotFull;
    int P, Ntot, chunks_per_split, nchunks;
    int mbase, mend;                 // rows [mbase, mend) of D's channels handled by this launch
    int store;                       // one K slice: plain stores instead of atomics
    int nsplit, dbs, sbs, gbs;       // blockIdx.z = problem * nsplit + K slice; pointers advance by these strides per problem
    int dlat, Td, Hd, Wd, doff[3];   // dlat: D is a sub-lattice (Tq,Hq,Wq) at doff of a [N][Td][Hd][Wd] tensor instead of dense
    int mt, ntl;                     // tiles along Cd and along the columns; the grid is 1-D: mt * ntl * problems * slices blocks
    long long wss;                   // > 0: g is a workspace of K-slice images wss floats apart -- slice k leaves its partial sums in image k with
                                     // plain stores and the gradient re-layout adds the images in slice order (no atomics: bit-identical reruns)
};

// X6: the multiplications on the bf16 matrix cores (conv_x6.hip's scheme; both operands are activations, so both are split in registers: a
// lane's MFMA operand is eight consecutive positions of one column of the [position][column] tiles, read with eight ds_read_b32)
template <int BM, int BN, int ABL, int KB, bool X6 = false, bool HILO = true>
__device__ __forceinline__ void wgrad_body(const WgK& p, const int lid) {
    constexpr int BK = KB;                           // positions per chunk (shadows the file-level BK)
    constexpr int TM = BM / 64, TN = BN / 64;        // 2x2 waves
    static_assert(TM >= 1 && TN >= 1, "tile");
    __shared__ __attribute__((aligned(16))) float Ds[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Ss[2][BK][BN];
    __shared__ int4 ptab[3][BK];                     // per chunk: n, t0, h0, w0 of its 32 positions
    __shared__ int drow[3][BK];                      // row of D for each position (sub-lattice launches)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // lid: logical block id of this problem (XCD-aware order: the blocks of one K slice -- every tile reads the same rows of D, and
    // overlapping rows of S -- get consecutive ids, i.e. run on one XCD and share its L2)
    const int tiles = p.mt * p.ntl, tile = lid % tiles, bz = lid / tiles;
    const int m0 = p.mbase + (tile % p.mt) * BM, n0 = (tile / p.mt) * BN;
    // bz = problem * nsplit + K slice
    const int prob = bz / p.nsplit, slice = bz - prob * p.nsplit;
    const int c_begin = slice * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    if (c_begin >= c_end) return;
    const float* Dp = p.D + (size_t)prob * p.dbs;
    const float* Sp = p.S + (size_t)prob * p.sbs;
    float* gp = p.g + (size_t)prob * p.gbs + (size_t)slice * p.wss;

    // column decode for the S tile (constant over the K loop)
    constexpr int SC4 = BN / 4, DC4 = BM / 4;          // float4 columns per row
    constexpr int SRP = 256 / SC4, DRP = 256 / DC4;    // rows covered per pass
    const int scol = (tid % SC4) * 4, srow0 = tid / SC4;
    const int dcol = (tid % DC4) * 4, drow0 = tid / DC4;
    const int ncol = n0 + scol;
    const bool nval = ncol < p.Ntot;
    const int tap = nval ? ncol / p.Cs : 0, cs = ncol - tap * p.Cs;
    const int tapHW = p.ntap[1] * p.ntap[2];
    const int a_ = tap / tapHW, rem = tap - a_ * tapHW, b_ = rem / p.ntap[2], c_ = rem - b_ * p.ntap[2];
    const int dt = a_ * p.istep[0], dh = b_ * p.istep[1], dw = c_ * p.istep[2];
    const bool mval = (m0 + dcol) < p.mend;
    (void)rem;

    auto ptab_fill = [&](int c) {
        if (tid < BK) {
            const int pos = c * BK + tid;
            int4 info = make_int4(-1, 0, 0, 0);
            if (c < c_end && pos < p.P) {
                int m = pos;
                const int wq = m % p.Wq; m /= p.Wq;
                const int hq = m % p.Hq; m /= p.Hq;
                const int tq = m % p.Tq; const int n = m / p.Tq;
                info = make_int4(n, tq * p.istr[0] + p.ioff0[0], hq * p.istr[1] + p.ioff0[1],
                                 wq * p.istr[2] + p.ioff0[2]);
                drow[c % 3][tid] = p.dlat ? ((n * p.Td + tq + p.doff[0]) * p.Hd + hq + p.doff[1]) * p.Wd + wq + p.doff[2] : pos;
            }
            ptab[c % 3][tid] = info;
        }
    };
    constexpr int DN = BK / DRP, SN = BK / SRP;        // loads per thread
    // tiles go global -> LDS directly (global_load_lds_dwordx4): each wave-instruction writes 1 KiB = consecutive
    // float4 columns of consecutive position rows, which is exactly the lane-linear image the DMA produces
    // (LDS float offset of thread tid in pass j = (tid + 256*j) * 4); invalid rows / taps read a zero line.
    // LDS-DMA through buffer resources (glds16b): the bases are the two tensors (wave-uniform), a lane's offset is 32-bit -- one 32-bit
    // multiply-add per piece instead of a 64-bit one and an address select; invalid rows / taps are DMA_OOB lanes (zero-filled)
    const dma_rsrc_t rsD = dma_rsrc(Dp + m0), rsS = dma_rsrc(Sp);
    const unsigned dcolb = mval ? (unsigned)dcol * 4u : DMA_OOB, csb = (unsigned)cs * 4u;
    auto gload = [&](int c, int buf) {
        float* ld = &Ds[buf][0][0] + wave * 256;      // wave-uniform base; the DMA adds lane*16 B
        float* ls = &Ss[buf][0][0] + wave * 256;
        if (ABL != 5)
#pragma unroll
        for (int j = 0; j < DN; ++j) {
            const int r = drow0 + DRP * j;
            const int pos = c * BK + r;
            const bool v = mval && pos < p.P;
            glds16b(rsD, v ? (unsigned)drow[c % 3][r] * (unsigned)(p.ldd * 4) + dcolb : DMA_OOB, ld + j * 1024);
        }
        if (ABL != 4)
#pragma unroll
        for (int j = 0; j < SN; ++j) {
            const int r = srow0 + SRP * j;
            const int4 info = ptab[c % 3][r];
            const int t = info.y + dt, h = info.z + dh, w = info.w + dw;
            const bool v = nval && info.x >= 0 && (unsigned)t < (unsigned)p.Ts && (unsigned)h < (unsigned)p.Hs &&
                           (unsigned)w < (unsigned)p.Ws;
            const unsigned ps = (unsigned)(((info.x * p.Ts + t) * p.Hs + h) * p.Ws + w);
            glds16b(rsS, v ? ps * (unsigned)(p.lds * 4) + csb : DMA_OOB, ls + j * 1024);
        }
    };

    f32x16 acc[TM][TN];
    f32x16 acc_lo[(X6 && HILO) ? TM : 1][(X6 && HILO) ? TN : 1];          // X6: `acc` takes the h*h products, acc_lo the five small ones (conv_x6.hip)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if constexpr (X6 && HILO) acc_lo[i][j][r] = 0.f; }

    ptab_fill(c_begin);
    ptab_fill(c_begin + 1);
    __syncthreads();
    gload(c_begin, 0);
    PC_SYNC_DMA();
    const int ml = wm * (BM / 2) + (lane & 31), nl = wn * (BN / 2) + (lane & 31), kh = lane >> 5;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (ABL >= 1 && ABL <= 3) ? 0 : ((c - c_begin) & 1);
        if (!ABL || ABL >= 4) ptab_fill(c + 2);
        if ((!ABL || ABL == 4 || ABL == 5) && c + 1 < c_end) gload(c + 1, buf ^ 1);   // 4: D tile only, 5: S tile only, 6: position table only
        if constexpr (X6) {
            static_assert(BK % 16 == 0 && ABL == 0, "whole k16 steps");
            typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
            typedef __attribute__((ext_vector_type(4))) uint32_t u4;
            auto load_split = [&](int s16, u4 (&A)[TM][3], u4 (&B)[TN][3]) {
                const int r0 = 16 * s16 + 8 * kh;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = Ds[buf][r0 + q][ml + i * 32];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { uint32_t h_, m_, l_; x6_split2(v[2 * q], v[2 * q + 1], h_, m_, l_); A[i][0][q] = h_; A[i][1][q] = m_; A[i][2][q] = l_; }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = Ss[buf][r0 + q][nl + j * 32];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { uint32_t h_, m_, l_; x6_split2(v[2 * q], v[2 * q + 1], h_, m_, l_); B[j][0][q] = h_; B[j][1][q] = m_; B[j][2][q] = l_; }
                }
            };
#define WG_MF(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, X), __builtin_bit_cast(bf8, Y), Cc, 0, 0, 0)
            auto mma = [&](const u4 (&A)[TM][3], const u4 (&B)[TN][3]) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16& lo = HILO ? acc_lo[HILO ? i : 0][HILO ? j : 0] : acc[i][j];
                        WG_MF(A[i][0], B[j][2], lo); WG_MF(A[i][2], B[j][0], lo); WG_MF(A[i][1], B[j][1], lo);
                        WG_MF(A[i][0], B[j][1], lo); WG_MF(A[i][1], B[j][0], lo);
                        WG_MF(A[i][0], B[j][0], acc[i][j]);
                    }
            };
            // (issuing the next step's reads and split in front of this step's MFMAs from a second register set, as wgrad3_x6_kernel does, was
            // measured on the 64 x 128 tile and bought nothing: 0.734 against 0.737 ms per step over its 19 launches -- two blocks per CU already
            // cover the split's latency)
#pragma unroll
            for (int s16 = 0; s16 < BK / 16; ++s16) {
                u4 A[TM][3], B[TN][3];
                load_split(s16, A, B);
                mma(A, B);
            }
#undef WG_MF
        } else
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Ds[buf][(ABL >= 2 ? 0 : ks * 2) + kh][ml + i * 32];     // ABL >= 2: one read for the chunk
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Ss[buf][(ABL >= 2 ? 0 : ks * 2) + kh][nl + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (ABL < 3) PC_SYNC_DMA();        // the LDS-DMA of chunk c+1 has landed and the reads of chunk c are fenced
    }
    if constexpr (X6 && HILO) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += acc_lo[i][j];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.mend) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
                if (n < p.Ntot) {
                    // trimmed column (tap_local, cs) -> column of the full [KT*KH*KW][Cs] layout
                    const int tl = n / p.Cs, cc = n - tl * p.Cs;
                    const int ta = tl / tapHW, tr = tl - ta * tapHW, tb = tr / p.ntap[2], tc = tr - tb * p.ntap[2];
                    const int full = ((p.wk0[0] + ta) * p.KH + p.wk0[1] + tb) * p.KW + p.wk0[2] + tc;
                    float* dst = gp + (size_t)m * p.NtotFull + (size_t)full * p.Cs + cc;
                    if (p.store || p.wss) *dst = acc[i][j][r];
                    else atomicAdd(dst, acc[i][j][r]);
                }
            }
        }
}

template <int BM, int BN, int ABL = 0, int KB = 32>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgK p) {
    wgrad_body<BM, BN, ABL, KB>(p, xcd_remap(blockIdx.x, gridDim.x));
}
template <int BM, int BN, bool HILO = true>
__global__ __launch_bounds__(256, 2) void wgrad_x6_kernel(const WgK p) {
    wgrad_body<BM, BN, 0, 32, true, HILO>(p, xcd_remap(blockIdx.x, gridDim.x));
}

// Several weight-gradient problems in ONE grid (pc_conv_wgrad_multi): the wgrads of one Inception module, or the eight position
// classes of the merged tail.  Each problem keeps its own descriptor (by value in the kernel arguments) and a contiguous range of
// blocks sized in proportion to its work, so split-K fills the chip once per group instead of once per layer: fewer, longer K
// slices (a slice's 64 KB of atomics buy 4x the FLOPs) and no 30-microsecond launches for 1 GFLOP of work.
constexpr int WG_MAXJOBS = 8;
struct WgMulti { int njobs; int first[WG_MAXJOBS + 1]; WgK job[WG_MAXJOBS]; };

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wgrad_multi_kernel(const WgMulti m) {
    int j = 0;
    const int bid = blockIdx.x;
    while (j + 1 < m.njobs && bid >= m.first[j + 1]) ++j;
    wgrad_body<BM, BN, 0, 32>(m.job[j], xcd_remap(bid - m.first[j], m.first[j + 1] - m.first[j]));
}


// Weight gradient of a conv with KW = 3 (padding 1) or 9 (no padding: the spectral forms' taps) taps of stride 1 along w
// (any stride along t / h) and Cs % 32 == 0: the K chunk is a segment of ONE image row, the column tile is (kw = 0..2) x 64 (or 32) channels of one (kt, kh)
// tap pair, and the KW taps read the SAME LDS tile of BKP + KW - 1 input positions at row offsets 0..KW-1 -- the
// gathered operand is fetched once per KW taps.  The generic kernel above is bound by the LDS-DMA fill of its two
// streaming operands; for the 64-channel layers this cuts the fill per FLOP by 1.8x.
struct Wg3K {
    const float* D; const float* S; float* g;
    int N, T, H, W, Cd, ldd, Cs, lds;
    int ntap_t, ntap_h, wk0_t, wk0_h, KH;     // (kt, kh) taps present (trimmed) and their place in the full [KT][KH][3] layout
    int Ts, Hs, istr_t, istr_h, ioff_t, ioff_h; // S row of output (t, h) and local tap (a, b): (t*istr_t + ioff_t + a, h*istr_h + ioff_h + b)
    int nseg, nchunks, chunks_per_split, nsplit, mt, ncs;   // segments per row; K chunks = N*T*H*nseg
    int taps_full;                            // KT*KH*KW: g is [Cd][taps_full][Cs]
    int Wsw, padw;                            // width of S and the padding along w (D width W = Wsw - KW + 1 + 2*padw)
    int nprob, store; long long dbs, sbs, gbs; // independent problems in one launch (pointer strides); plain stores (one K slice)
    long long wss;                            // > 0: K-slice images in a workspace instead of atomics (WgK::wss)
};

template <int BM, int BKP, int CSB, int WMW, int KW = 3>
__global__ __launch_bounds__(256, 2) void wgrad3_kernel(const Wg3K p) {
    constexpr int BN = KW * CSB, WNW = 4 / WMW;                            // WMW x WNW waves
    constexpr int TM = BM / WMW / 32, TN = BN / WNW / 32;
    static_assert(BM % (WMW * 32) == 0 && BN % (WNW * 32) == 0, "wave tiles");
    constexpr int SPIECE = 256 / CSB;                                      // S rows per 1 KiB DMA piece
    constexpr int SROWS = (BKP + KW - 1 + SPIECE - 1) / SPIECE * SPIECE;   // S tile rows padded to whole DMA pieces
    constexpr int DI = BKP * BM * 4 / 1024, SI = SROWS * CSB * 4 / 1024;   // DMA wave-instructions per tile
    static_assert((BKP * BM * 4) % 1024 == 0, "D tile must be whole DMA pieces");
    // One LDS variable per ring slot and the slot a compile-time constant (the chunk loop is unrolled by two): the compiler's
    // s_waitcnt pass orders every LDS read behind every in-flight LDS-DMA write it cannot prove disjoint, and it can only prove it
    // for distinct variables.  With Ds[2][..] indexed by a run-time slot it put s_waitcnt vmcnt(0) between the fetch of chunk c + 1
    // and the first operand read of chunk c -- the fetch was never in flight during the MFMA loop (MFMA busy 0.61).
    __shared__ __attribute__((aligned(16))) float Ds0[BKP][BM];
    __shared__ __attribute__((aligned(16))) float Ds1[BKP][BM];
    __shared__ __attribute__((aligned(16))) float Ss0[SROWS][CSB];
    __shared__ __attribute__((aligned(16))) float Ss1[SROWS][CSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.mt * p.ncs * p.ntap_t * p.ntap_h;
    int tile = lid % tiles;
    const int slice = (lid / tiles) % p.nsplit, prob = lid / (tiles * p.nsplit);
    const float* Dp = p.D + (size_t)prob * p.dbs;
    const float* Sp = p.S + (size_t)prob * p.sbs;
    float* gp = p.g + (size_t)prob * p.gbs + (size_t)slice * p.wss;
    const int mtile = tile % p.mt; tile /= p.mt;
    const int csb = tile % p.ncs; tile /= p.ncs;
    const int kh_ = tile % p.ntap_h, kt_ = tile / p.ntap_h;
    const int m0 = mtile * BM, cs0 = csb * CSB;
    const int c_begin = slice * p.chunks_per_split, c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    if (c_begin >= c_end) return;

    // chunk -> (n, t, h, seg), kept as a counter that advances with every fetch (chunks are fetched in order: no divisions in the
    // K loop); a fetch returns false when the tap pair reads outside the volume (the chunk contributes nothing)
    int q_seg = c_begin % p.nseg, q_h, q_t, q_n;
    { int r = c_begin / p.nseg; q_h = r % p.H; r /= p.H; q_t = r % p.T; q_n = r / p.T; }
    auto decode = [&](int& row_d, int& row_s, int& w0) -> bool {
        w0 = q_seg * BKP;
        row_d = ((q_n * p.T + q_t) * p.H + q_h) * p.W;
        const int ts = q_t * p.istr_t + p.ioff_t + kt_, hs = q_h * p.istr_h + p.ioff_h + kh_;
        row_s = ((q_n * p.Ts + ts) * p.Hs + hs) * p.Wsw;
        if (++q_seg == p.nseg) { q_seg = 0; if (++q_h == p.H) { q_h = 0; if (++q_t == p.T) { q_t = 0; ++q_n; } } }
        return (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs;
    };
    auto gload = [&](auto slot) -> bool {
        constexpr int buf = decltype(slot)::value;
        int row_d, row_s, w0;
        if (!decode(row_d, row_s, w0)) return false;
        float* ld = buf ? &Ds1[0][0] : &Ds0[0][0];
        float* ls = buf ? &Ss1[0][0] : &Ss0[0][0];
        // wave-uniform bases: the segment's first position (the S base may lie padw positions in front of its row: only address arithmetic)
        const dma_rsrc_t rsD = dma_rsrc(Dp + (size_t)(row_d + w0) * p.ldd + m0);
        const dma_rsrc_t rsS = dma_rsrc(Sp + ((long long)row_s + w0 - p.padw) * p.lds + cs0);
#pragma unroll
        for (int jj = 0; jj < (DI + 3) / 4; ++jj) {                        // D tile: BKP rows x BM channels
            const int i = jj * 4 + wave;
            if (i >= DI) break;
            const int e = i * 64 + lane, r = e / (BM / 4), c4 = e % (BM / 4);
            const bool v = (w0 + r) < p.W && (m0 + c4 * 4) < p.Cd;
            glds16b(rsD, v ? (unsigned)(r * p.ldd + c4 * 4) * 4u : DMA_OOB, ld + i * 256);
        }
#pragma unroll
        for (int jj = 0; jj < (SI + 3) / 4; ++jj) {                        // S tile: positions w0-pad .. w0-pad+BKP+KW-2 (+ padding rows)
            const int i = jj * 4 + wave;
            if (i >= SI) break;
            const int e = i * 64 + lane, r = e / (CSB / 4), c4 = e % (CSB / 4);
            const int w = w0 - p.padw + r;
            const bool v = r < BKP + KW - 1 && (unsigned)w < (unsigned)p.Wsw;
            glds16b(rsS, v ? (unsigned)(r * p.lds + c4 * 4) * 4u : DMA_OOB, ls + i * 256);
        }
        return true;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // column tile j of wave wn: columns wn*96 + 32j .. +31 of (kw, cs): kw = col / 64, cs = col % 64
    int kwj[TN], csj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int col = wn * (BN / WNW) + 32 * j; kwj[j] = col / CSB; csj[j] = col % CSB + (lane & 31); }
    const int ml = wm * (BM / WMW) + (lane & 31), kh = lane >> 5;
    bool live = gload(std::integral_constant<int, 0>{});
    PC_SYNC_DMA();
    auto chunk = [&](auto slot, int c) {
        constexpr int buf = decltype(slot)::value;
        const bool next_live = c + 1 < c_end ? gload(std::integral_constant<int, buf ^ 1>{}) : false;
        if (live) {
            const float (*Dt)[BM] = buf ? Ds1 : Ds0;
            const float (*St)[CSB] = buf ? Ss1 : Ss0;
#pragma unroll
            for (int ks = 0; ks < BKP / 2; ++ks) {
                const int pp = ks * 2 + kh;
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = Dt[pp][ml + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = St[pp + kwj[j]][csj[j]];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        PC_SYNC_DMA();        // the LDS-DMA of chunk c+1 has landed and the reads of chunk c are fenced
        live = next_live;
    };
    for (int c = c_begin; c < c_end; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < c_end) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    const int tapbase = ((kt_ + p.wk0_t) * p.KH + kh_ + p.wk0_h) * KW;    // + kw
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (BM / WMW) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.Cd) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* dst = gp + ((size_t)m * p.taps_full + tapbase + kwj[j]) * p.Cs + cs0 + csj[j];
                if (p.store || p.wss) *dst = acc[i][j][r];
                else atomicAdd(dst, acc[i][j][r]);
            }
        }
}



// ---- the same row-segment weight gradient with the multiplications on the bf16 matrix cores (csrc/conv_x6.hip has the scheme: three bf16
// terms per fp32 operand, six products, hi / lo accumulators).  Both operands are activations here, so both are split in registers: a lane's
// operand of v_mfma_f32_32x32x16_bf16 is eight consecutive POSITIONS of one channel -- eight ds_read_b32 down a column of the [position]
// [channel] tile the LDS-DMA left (conflict-free: a half-wave reads 32 consecutive channels of one row) -- and the KW = 3 taps share them:
// the ten source values at positions 8 g .. 8 g + 9 are read and split ONCE, tap 0 takes the packed pairs 0..3, tap 2 the pairs 1..4, tap 1
// re-pairs them with four v_alignbit per plane.  Per k16 step and wave: 18 reads + 111 vector instructions beside 18 MFMAs.
// One 32-channel tile of D and one 32-channel half of S per wave (all three taps): BM = 32 WMW, CSB = 32 (4 / WMW).  BKP (a multiple of
// 16) positions per chunk; a segment that runs past the row end reads zeros (LDS-DMA out-of-range lanes).
template <int BM, int BKP, int CSB, int WMW>
__global__ __launch_bounds__(256, 2) void wgrad3_x6_kernel(const Wg3K p) {
    constexpr int KW = 3, WNW = 4 / WMW, NS = BKP / 16;
    static_assert(BM == 32 * WMW && CSB == 32 * WNW && BKP % 16 == 0, "one 32-channel tile of D and of S per wave");
    constexpr int SPIECE = 256 / CSB;
    constexpr int SROWS = (BKP + KW - 1 + SPIECE - 1) / SPIECE * SPIECE;
    constexpr int DI = BKP * BM * 4 / 1024, SI = SROWS * CSB * 4 / 1024;
    static_assert((BKP * BM * 4) % 1024 == 0, "D tile must be whole DMA pieces");
    __shared__ __attribute__((aligned(16))) float Ds0[BKP][BM];          // one variable per ring slot: see wgrad3_kernel
    __shared__ __attribute__((aligned(16))) float Ds1[BKP][BM];
    __shared__ __attribute__((aligned(16))) float Ss0[SROWS][CSB];
    __shared__ __attribute__((aligned(16))) float Ss1[SROWS][CSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.mt * p.ncs * p.ntap_t * p.ntap_h;
    int tile = lid % tiles;
    const int slice = (lid / tiles) % p.nsplit, prob = lid / (tiles * p.nsplit);
    const float* Dp = p.D + (size_t)prob * p.dbs;
    const float* Sp = p.S + (size_t)prob * p.sbs;
    float* gp = p.g + (size_t)prob * p.gbs + (size_t)slice * p.wss;
    const int mtile = tile % p.mt; tile /= p.mt;
    const int csb = tile % p.ncs; tile /= p.ncs;
    const int kh_ = tile % p.ntap_h, kt_ = tile / p.ntap_h;
    const int m0 = mtile * BM, cs0 = csb * CSB;
    const int c_begin = slice * p.chunks_per_split, c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    if (c_begin >= c_end) return;

    int q_seg = c_begin % p.nseg, q_h, q_t, q_n;
    { int r = c_begin / p.nseg; q_h = r % p.H; r /= p.H; q_t = r % p.T; q_n = r / p.T; }
    auto decode = [&](int& row_d, int& row_s, int& w0) -> bool {
        w0 = q_seg * BKP;
        row_d = ((q_n * p.T + q_t) * p.H + q_h) * p.W;
        const int ts = q_t * p.istr_t + p.ioff_t + kt_, hs = q_h * p.istr_h + p.ioff_h + kh_;
        row_s = ((q_n * p.Ts + ts) * p.Hs + hs) * p.Wsw;
        if (++q_seg == p.nseg) { q_seg = 0; if (++q_h == p.H) { q_h = 0; if (++q_t == p.T) { q_t = 0; ++q_n; } } }
        return (unsigned)ts < (unsigned)p.Ts && (unsigned)hs < (unsigned)p.Hs;
    };
    auto gload = [&](auto slot) -> bool {
        constexpr int buf = decltype(slot)::value;
        int row_d, row_s, w0;
        if (!decode(row_d, row_s, w0)) return false;
        float* ld = buf ? &Ds1[0][0] : &Ds0[0][0];
        float* ls = buf ? &Ss1[0][0] : &Ss0[0][0];
        const dma_rsrc_t rsD = dma_rsrc(Dp + (size_t)(row_d + w0) * p.ldd + m0);
        const dma_rsrc_t rsS = dma_rsrc(Sp + ((long long)row_s + w0 - p.padw) * p.lds + cs0);
#pragma unroll
        for (int jj = 0; jj < (DI + 3) / 4; ++jj) {
            const int i = jj * 4 + wave;
            if (i >= DI) break;
            const int e = i * 64 + lane, r = e / (BM / 4), c4 = e % (BM / 4);
            const bool v = (w0 + r) < p.W && (m0 + c4 * 4) < p.Cd;                   // past the row end: zeros
            glds16b(rsD, v ? (unsigned)(r * p.ldd + c4 * 4) * 4u : DMA_OOB, ld + i * 256);
        }
#pragma unroll
        for (int jj = 0; jj < (SI + 3) / 4; ++jj) {
            const int i = jj * 4 + wave;
            if (i >= SI) break;
            const int e = i * 64 + lane, r = e / (CSB / 4), c4 = e % (CSB / 4);
            const int w = w0 - p.padw + r;
            const bool v = r < BKP + KW - 1 && (unsigned)w < (unsigned)p.Wsw;
            glds16b(rsS, v ? (unsigned)(r * p.lds + c4 * 4) * 4u : DMA_OOB, ls + i * 256);
        }
        return true;
    };

    f32x16 acc_hi[KW], acc_lo[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_hi[j][r] = 0.f; acc_lo[j][r] = 0.f; }
    const int ml = wm * 32 + (lane & 31), csl = wn * 32 + (lane & 31), kg = lane >> 5;
    typedef uint32_t u32;
    // planes of one k16 step: A (D^T) [plane][4 regs]; B pairs p = 0..4 of the ten source values [plane][5]
    auto load_split = [&](const float (*Dt)[BM], const float (*St)[CSB], int s, u32 (&pa)[3][4], u32 (&pb)[3][5]) {
        const int r0 = 16 * s + 8 * kg;
        float a[8], b[10];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = Dt[r0 + j][ml];
#pragma unroll
        for (int j = 0; j < 10; ++j) b[j] = St[r0 + j][csl];
#pragma unroll
        for (int q = 0; q < 4; ++q) x6_split2(a[2 * q], a[2 * q + 1], pa[0][q], pa[1][q], pa[2][q]);
#pragma unroll
        for (int q = 0; q < 5; ++q) x6_split2(b[2 * q], b[2 * q + 1], pb[0][q], pb[1][q], pb[2][q]);
    };
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    auto mma_step = [&](const u32 (&pa)[3][4], const u32 (&pb)[3][5]) {
        u4 A[3], B[KW][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            A[pl] = u4{pa[pl][0], pa[pl][1], pa[pl][2], pa[pl][3]};
            B[0][pl] = u4{pb[pl][0], pb[pl][1], pb[pl][2], pb[pl][3]};
            B[2][pl] = u4{pb[pl][1], pb[pl][2], pb[pl][3], pb[pl][4]};
            // tap 1: positions 1..8 = (hi of pair q, lo of pair q + 1)
            B[1][pl] = u4{__builtin_amdgcn_alignbit(pb[pl][1], pb[pl][0], 16), __builtin_amdgcn_alignbit(pb[pl][2], pb[pl][1], 16),
                          __builtin_amdgcn_alignbit(pb[pl][3], pb[pl][2], 16), __builtin_amdgcn_alignbit(pb[pl][4], pb[pl][3], 16)};
        }
#define WG_MF(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, X), __builtin_bit_cast(bf8, Y), Cc, 0, 0, 0)
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            WG_MF(A[0], B[j][2], acc_lo[j]); WG_MF(A[2], B[j][0], acc_lo[j]); WG_MF(A[1], B[j][1], acc_lo[j]);
            WG_MF(A[0], B[j][1], acc_lo[j]); WG_MF(A[1], B[j][0], acc_lo[j]);
            WG_MF(A[0], B[j][0], acc_hi[j]);
        }
#undef WG_MF
    };
    bool live = gload(std::integral_constant<int, 0>{});
    PC_SYNC_DMA();
    auto chunk = [&](auto slot, int c) {
        constexpr int buf = decltype(slot)::value;
        const bool next_live = c + 1 < c_end ? gload(std::integral_constant<int, buf ^ 1>{}) : false;
        if (live) {
            const float (*Dt)[BM] = buf ? Ds1 : Ds0;
            const float (*St)[CSB] = buf ? Ss1 : Ss0;
            u32 pa[2][3][4], pb[2][3][5];
            load_split(Dt, St, 0, pa[0], pb[0]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS) load_split(Dt, St, s + 1, pa[(s + 1) & 1], pb[(s + 1) & 1]);      // the next step's reads and split beside this step's MFMAs
                mma_step(pa[s & 1], pb[s & 1]);
            }
        }
        PC_SYNC_DMA();        // the LDS-DMA of chunk c+1 has landed and the reads of chunk c are fenced
        live = next_live;
    };
    for (int c = c_begin; c < c_end; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < c_end) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    const int tapbase = ((kt_ + p.wk0_t) * p.KH + kh_ + p.wk0_h) * KW;    // + kw
#pragma unroll
    for (int j = 0; j < KW; ++j) {
        const f32x16 acc = acc_hi[j] + acc_lo[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.Cd) continue;
            float* dst = gp + ((size_t)m * p.taps_full + tapbase + j) * p.Cs + cs0 + csl;
            if (p.store || p.wss) *dst = acc[r];
            else atomicAdd(dst, acc[r]);
        }
    }
}


// ---- weight gradient of a 4-channel source with stride SW along w (the stem: 7x7x7, stride 2, 3 channels padded to 4) --------
// One tap of 4 channels is one 16-byte piece, so the generic kernel gathers its S tile piece by piece (64 taps x 16 positions
// = 1024 pieces per chunk, each input piece fetched ~3x per chunk).  Here the K chunk is a segment of BKP positions of ONE
// output row, and a block owns ALL (kh, kw) taps of one kt: per kh one LDS row of SW*BKP + KW - 1 input pieces, fetched once.
// Column n = kw*4 + cs of tap row kh for position k is LDS dword kh*ROW + 4*SW*k + n: 32 consecutive dwords per MFMA operand,
// conflict-free.  (kw = KW..7 are padding columns, dropped at the store.)  Waves: 2 halves of the 64 channels of D x 2 k-step
// parities, 7 (= KH) 32x32 accumulators each.  Chunks whose kt tap reads outside the volume along t are not enumerated at all,
// and the K slices are dealt to the kt taps in proportion to their valid chunks.
struct Wg4K {
    const float* D; const float* S; float* g;
    int N, T, H, W, Cd, ldd, lds;
    int ntap_t, wk0_t, wk0_h, KH;
    int Ts, Hs, Wsw, istr_t, istr_h, ioff_t, ioff_h, padw;
    int nseg, mt, taps_full;
    int pre[11];                               // slice prefix per kt (ntap_t + 1 entries)
    int interleave;                            // 1: order[] below is valid
    long long wss;                             // > 0: K-slice images in a workspace instead of atomics (WgK::wss); image index = slice within its kt
    unsigned short order[1024];                // block q of an m tile -> slice index (kt slices interleaved by their position in the volume)
};

// PACK3 (PC_WG_CS3: the 4th source channel is padding): the NKH * KW * 3 real columns are packed into ceil(147 / 32) = 5
// accumulators instead of one 32-column accumulator per kh with 21 real columns in it -- 5 MFMAs per k-step instead of 7.  Column
// G = jj * 32 + lane -> (kh, kw, cs) = (G / 21, G % 21 / 3, G % 3) reads LDS dword kh*ROW + 4*SW*k + 4*kw + cs: per-lane constant
// offsets; the padding dwords leave a quarter of the banks unused, so these ds_read_b32 are 2-way conflicted (2 of 22 LDS cycles
// per k-step more per wave, far from the LDS limit).
template <int BKP, int KW, int SW, int NKH, bool PACK3>
__global__ __launch_bounds__(256, 2) void wgrad4_kernel(const Wg4K p) {
    constexpr int BM = 64;
    constexpr int ROWP = SW * BKP + KW - 1;                  // pieces per kh row
    static_assert(SW * (BKP - 1) + 7 < ROWP + 1, "padding columns stay inside the row");
    constexpr int SPIECES = NKH * ROWP, SI = (SPIECES + 63) / 64, DI = BKP * BM * 4 / 1024;
    static_assert((BKP * BM * 4) % 1024 == 0 && (BKP / 2) % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) float Ds0[BKP][BM];          // one variable per ring slot: see wgrad3_kernel
    __shared__ __attribute__((aligned(16))) float Ds1[BKP][BM];
    __shared__ __attribute__((aligned(16))) float Ss0[SI * 256];
    __shared__ __attribute__((aligned(16))) float Ss1[SI * 256];
    __shared__ float xch[2][16][64];                                     // epilogue of the K-slice-image form: wave wk = 1 -> wave wk = 0
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wk = wave >> 1;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // The slices of the seven kt taps that cover the same stretch of the volume read the same D rows: they sit next to each other in block order
    // (one XCD, one moment), so those rows come out of L2 instead of HBM seven times (1.6 GB per launch for 0.31 GB of operands before)
    const int mtile = lid % p.mt, sl = p.interleave ? (int)p.order[lid / p.mt] : lid / p.mt;
    int kt_ = 0;
    while (kt_ + 1 < p.ntap_t && sl >= p.pre[kt_ + 1]) ++kt_;
    const int slice = sl - p.pre[kt_], nsplit = p.pre[kt_ + 1] - p.pre[kt_];
    // output t with a valid source t for this tap
    int tlo = 0, thi = p.T - 1;
    while (tlo <= thi && tlo * p.istr_t + p.ioff_t + kt_ < 0) ++tlo;
    while (thi >= tlo && thi * p.istr_t + p.ioff_t + kt_ >= p.Ts) --thi;
    const int ntv = thi - tlo + 1;
    const int nchunks = p.N * ntv * p.H * p.nseg;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = slice * cps, c_end = min(nchunks, c_begin + cps);
    if (c_begin >= c_end) return;
    const int m0 = mtile * BM;

    // chunk -> (n, t, h, seg) as a counter advanced by every fetch (chunks are fetched in order)
    int q_seg = c_begin % p.nseg, q_h, q_t, q_n;
    { int r = c_begin / p.nseg; q_h = r % p.H; r /= p.H; q_t = r % ntv; q_n = r / ntv; }
    auto gload = [&](auto slot) {
        constexpr int buf = decltype(slot)::value;
        const int seg = q_seg, h = q_h, t = tlo + q_t, n = q_n;
        if (++q_seg == p.nseg) { q_seg = 0; if (++q_h == p.H) { q_h = 0; if (++q_t == ntv) { q_t = 0; ++q_n; } } }
        const int w0 = seg * BKP;
        const int row_d = ((n * p.T + t) * p.H + h) * p.W;
        const int ts = t * p.istr_t + p.ioff_t + kt_, hs0 = h * p.istr_h + p.ioff_h;
        float* ld = buf ? &Ds1[0][0] : &Ds0[0][0];
        float* ls = buf ? &Ss1[0] : &Ss0[0];
        const dma_rsrc_t rsD = dma_rsrc(p.D + (size_t)(row_d + w0) * p.ldd + m0);
        // S rows hs0 .. hs0 + NKH - 1 of frame (n, ts), from w0*SW - padw on: base at (hs0, w0*SW - padw) -- possibly in front of the frame
        const dma_rsrc_t rsS = dma_rsrc(p.S + (((long long)(n * p.Ts + ts) * p.Hs + hs0) * p.Wsw + (long long)w0 * SW - p.padw) * p.lds);
#pragma unroll
        for (int jj = 0; jj < (DI + 3) / 4; ++jj) {
            const int i = jj * 4 + wave;
            if (i >= DI) break;
            const int e = i * 64 + lane, rr = e / (BM / 4), c4 = e % (BM / 4);
            const bool v = (m0 + c4 * 4) < p.Cd;
            glds16b(rsD, v ? (unsigned)(rr * p.ldd + c4 * 4) * 4u : DMA_OOB, ld + i * 256);
        }
#pragma unroll
        for (int jj = 0; jj < (SI + 3) / 4; ++jj) {
            const int i = jj * 4 + wave;
            if (i >= SI) break;
            const int e = i * 64 + lane, kh = e / ROWP, rr = e - kh * ROWP;
            const int hs = hs0 + kh, w = w0 * SW - p.padw + rr;
            const bool v = kh < NKH && (unsigned)hs < (unsigned)p.Hs && (unsigned)w < (unsigned)p.Wsw;
            glds16b(rsS, v ? (unsigned)((kh * p.Wsw + rr) * p.lds) * 4u : DMA_OOB, ls + i * 256);     
        }
    };

    constexpr int NCOL = NKH * KW * 3, NACC = PACK3 ? (NCOL + 31) / 32 : NKH;
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nl = lane & 31, kh2 = lane >> 5, ml = wm * 32 + nl;
    int boff[NACC], gcol[NACC];                // LDS dword of this lane's column in accumulator j / its offset inside g's [tap][4] rows (-1: none)
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        if constexpr (PACK3) {
            const int G = j * 32 + nl, kh = G / (KW * 3), rem = G - kh * (KW * 3), kw = rem / 3, cs = rem - kw * 3;
            const bool v = G < NCOL;
            boff[j] = v ? kh * ROWP * 4 + kw * 4 + cs : 0;
            gcol[j] = v ? (kh * KW + kw) * 4 + cs : -1;
        } else {
            boff[j] = j * ROWP * 4 + nl;
            gcol[j] = nl < KW * 4 ? j * KW * 4 + nl : -1;          // kw = KW..7 are padding columns
        }
    }
    gload(std::integral_constant<int, 0>{});
    PC_SYNC_DMA();
    auto chunk = [&](auto slot, int c) {
        constexpr int buf = decltype(slot)::value;
        if (c + 1 < c_end) gload(std::integral_constant<int, buf ^ 1>{});
        const float* sb = buf ? &Ss1[0] : &Ss0[0];
        const float (*Dt)[BM] = buf ? Ds1 : Ds0;
#pragma unroll
        for (int q = 0; q < BKP / 4; ++q) {
            const int pp = (2 * q + wk) * 2 + kh2;
            const float af = Dt[pp][ml];
            float bf[NACC];
#pragma unroll
            for (int j = 0; j < NACC; ++j) bf[j] = sb[boff[j] + 4 * SW * pp];
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[j], acc[j], 0, 0, 0);
        }
        PC_SYNC_DMA();
    };
    for (int c = c_begin; c < c_end; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < c_end) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    const size_t tap0 = (size_t)((kt_ + p.wk0_t) * p.KH + p.wk0_h) * KW * 4;       // g offset of (kt, kh = 0, kw = 0, cs = 0)
    if (p.wss) {
        // K-slice image instead of atomics: the two k-step parities (waves wk = 0 / 1) hold partial sums of the SAME outputs, which the atomics
        // used to combine -- here wave wk = 1 hands its accumulators to wave wk = 0 through LDS (one accumulator per round), which adds them in a
        // fixed order and stores the image with plain stores
        float* image = p.g + (size_t)slice * p.wss;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (wk == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[wm][r][lane] = acc[j][r];
            }
            __syncthreads();
            if (wk == 0 && gcol[j] >= 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.Cd) image[(size_t)m * p.taps_full * 4 + tap0 + gcol[j]] = acc[j][r] + xch[wm][r][lane];
                }
            }
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.Cd) continue;
#pragma unroll
        for (int j = 0; j < NACC; ++j)
            if (gcol[j] >= 0) atomicAdd(p.g + (size_t)m * p.taps_full * 4 + tap0 + gcol[j], acc[j][r]);
    }
}

// ---- the stem's weight gradient on the bf16 matrix cores (round 6; conv_x6.hip has the scheme: three bf16 terms per fp32 operand, six products,
// hi / lo accumulators).  Same problem and LDS images as wgrad4_kernel<.., PACK3> -- a block owns all 7 x 7 (kh, kw) taps x 3 real source channels
// of one kt, packed into five 32-column accumulators; every input piece is fetched once per chunk -- but the K chunk is TWO sub-chunks of 16
// consecutive output positions of a row (112 = 7 x 16: no padding positions, where 28-position segments would leave 4 of 32 empty), one k16
// step each: waves = 2 halves of D's 64 channels x the 2 sub-chunks.  Both operands are activations, so both are split in registers: a lane's
// MFMA operand is eight consecutive POSITIONS of one column -- eight ds_read_b32 down the D tile, eight ds_read_b32 eight dwords apart (SW * 4)
// along an input row for (kh, kw, cs).  Per wave and chunk: 48 LDS reads, 24 two-element splits, 30 MFMAs -- 0.6 microseconds of matrix-pipe
// time, less than the latency of an LDS-DMA fetch: the tiles live in a THREE-deep ring (chunk c + 2 is fetched while chunk c is multiplied) with
// counted waits (every wave issues the same five DMA instructions per chunk, so "chunk c has landed" is vmcnt <= 5 while chunk c + 1 is in flight).
template <int KW, int SW, int NKH>
__global__ __launch_bounds__(256, 2) void wgrad4_x6_kernel(const Wg4K p) {
    constexpr int BM = 64, SUB = 16, BKP = 2 * SUB;
    constexpr int ROWP = SW * SUB + KW - 1;                  // pieces per kh row of a sub-chunk
    constexpr int SPIECES = NKH * ROWP, SI = 6, SSUB = SI * 256, DI = BKP * BM * 4 / 1024;     // six DMA instructions per sub-chunk image (the last two partly / all padding)
    constexpr int NCOL = NKH * KW * 3, NACC = (NCOL + 31) / 32;
    constexpr int NDMA = 5;                                  // DMA instructions per wave and chunk: 2 of D, 3 of S
    static_assert(DI == 8 && SPIECES <= SI * 64 && 2 * SI == 12, "tile shape");
    __shared__ __attribute__((aligned(16))) float Ds0[BKP][BM];          // one variable per ring slot: see wgrad3_kernel
    __shared__ __attribute__((aligned(16))) float Ds1[BKP][BM];
    __shared__ __attribute__((aligned(16))) float Ds2[BKP][BM];
    __shared__ __attribute__((aligned(16))) float Ss0[2 * SSUB];
    __shared__ __attribute__((aligned(16))) float Ss1[2 * SSUB];
    __shared__ __attribute__((aligned(16))) float Ss2[2 * SSUB];
    __shared__ float xch[2][16][64];                                     // epilogue: wave wk = 1 -> wave wk = 0
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wk = wave >> 1;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = lid % p.mt, sl = p.interleave ? (int)p.order[lid / p.mt] : lid / p.mt;
    int kt_ = 0;
    while (kt_ + 1 < p.ntap_t && sl >= p.pre[kt_ + 1]) ++kt_;
    const int slice = sl - p.pre[kt_], nsplit = p.pre[kt_ + 1] - p.pre[kt_];
    int tlo = 0, thi = p.T - 1;
    while (tlo <= thi && tlo * p.istr_t + p.ioff_t + kt_ < 0) ++tlo;
    while (thi >= tlo && thi * p.istr_t + p.ioff_t + kt_ >= p.Ts) --thi;
    const int ntv = thi - tlo + 1;
    const int nsub = p.N * ntv * p.H * p.nseg;               // sub-chunks (p.nseg = ceil(W / 16) per row)
    const int nchunks = (nsub + 1) / 2;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = slice * cps, c_end = min(nchunks, c_begin + cps);
    if (c_begin >= c_end) return;
    const int m0 = mtile * BM;

    // sub-chunk -> (n, t, h, seg) as a counter advanced by every fetch (two sub-chunks per chunk, in order)
    int q_seg, q_h, q_t, q_n, q_idx = 2 * c_begin;
    { int r = q_idx; q_seg = r % p.nseg; r /= p.nseg; q_h = r % p.H; r /= p.H; q_t = r % ntv; q_n = r / ntv; }
    auto gload = [&](auto slot) {
        constexpr int buf = decltype(slot)::value;
        float* ld = buf == 0 ? &Ds0[0][0] : buf == 1 ? &Ds1[0][0] : &Ds2[0][0];
        float* ls = buf == 0 ? &Ss0[0] : buf == 1 ? &Ss1[0] : &Ss2[0];
        dma_rsrc_t rsS[2];
        int hs0_[2], w0_[2]; bool live_[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const bool live = q_idx < nsub;                   // an odd number of sub-chunks: the last chunk's second half reads zeros
            const int seg = q_seg, h = q_h, t = tlo + q_t, n = q_n;
            ++q_idx;
            if (++q_seg == p.nseg) { q_seg = 0; if (++q_h == p.H) { q_h = 0; if (++q_t == ntv) { q_t = 0; ++q_n; } } }
            const int w0 = seg * SUB;
            const int row_d = ((n * p.T + t) * p.H + h) * p.W;
            const int ts = t * p.istr_t + p.ioff_t + kt_, hs0 = h * p.istr_h + p.ioff_h;
            const dma_rsrc_t rsD = dma_rsrc(p.D + (live ? (size_t)(row_d + w0) * p.ldd + m0 : 0));
            rsS[sub] = dma_rsrc(p.S + (live ? (((long long)(n * p.Ts + ts) * p.Hs + hs0) * p.Wsw + (long long)w0 * SW - p.padw) * p.lds : 0));
            hs0_[sub] = hs0; w0_[sub] = w0; live_[sub] = live;
            {   // D: DMA instruction sub * 4 + wave = rows 16 sub + 4 wave .. + 3 of the chunk's tile
                const int i = sub * 4 + wave;
                const int rr = lane >> 4, c4 = lane & 15, r16 = wave * 4 + rr;
                const bool v = live && (w0 + r16) < p.W && (m0 + c4 * 4) < p.Cd;
                glds16b(rsD, v ? (unsigned)(r16 * p.ldd + c4 * 4) * 4u : DMA_OOB, ld + i * 256);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {                      // S: instruction wave + 4 jj of the chunk's twelve: sub-chunk image i / 6, its piece row i % 6
            const int i = jj * 4 + wave, sub = i / SI, ii = i - sub * SI;
            const int e = ii * 64 + lane, kh = e / ROWP, rr = e - kh * ROWP;
            const int hs = hs0_[sub] + kh, w = w0_[sub] * SW - p.padw + rr;
            const bool v = live_[sub] && kh < NKH && (unsigned)hs < (unsigned)p.Hs && (unsigned)w < (unsigned)p.Wsw;
            glds16b(rsS[sub], v ? (unsigned)((kh * p.Wsw + rr) * p.lds) * 4u : DMA_OOB, ls + i * 256);
        }
    };

    f32x16 acc_hi[NACC], acc_lo[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_hi[j][r] = 0.f; acc_lo[j][r] = 0.f; }
    const int nl = lane & 31, kg = lane >> 5, ml = wm * 32 + nl;
    int boff[NACC], gcol[NACC];                // LDS dword of this lane's column in accumulator j at its first position / its offset inside g's [tap][4] rows (-1: none)
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const int G = j * 32 + nl, kh = G / (KW * 3), rem = G - kh * (KW * 3), kw = rem / 3, cs = rem - kw * 3;
        const bool v = G < NCOL;
        boff[j] = (v ? kh * ROWP * 4 + kw * 4 + cs : 0) + wk * SSUB + 4 * SW * 8 * kg;
        gcol[j] = v ? (kh * KW + kw) * 4 + cs : -1;
    }
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
#define WG_MF(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, X), __builtin_bit_cast(bf8, Y), Cc, 0, 0, 0)
    auto chunk = [&](auto slot, int c) {
        constexpr int buf = decltype(slot)::value;
        // chunk c has landed for every wave (counted: chunk c + 1, if any, stays in flight) and every wave is done with chunk c - 1, whose slot
        // the fetch of chunk c + 2 overwrites.  The waits are written out (PC_SYNC_DMA, common.h: the compiler's own bookkeeping lost them once)
        if (c + 1 < c_end) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 2 < c_end) gload(std::integral_constant<int, (buf + 2) % 3>{});
        const float* sb = buf == 0 ? &Ss0[0] : buf == 1 ? &Ss1[0] : &Ss2[0];
        const float (*Dt)[BM] = buf == 0 ? Ds0 : buf == 1 ? Ds1 : Ds2;
        u4 A[3];
        {
            float a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = Dt[wk * SUB + 8 * kg + q][ml];
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint32_t h_, m_, l_; x6_split2(a[2 * q], a[2 * q + 1], h_, m_, l_); A[0][q] = h_; A[1][q] = m_; A[2][q] = l_; }
        }
        float b[NACC][8];
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) b[j][q] = sb[boff[j] + 4 * SW * q];
        // software pipeline inside the chunk: the split of column tile j + 1 is issued between the six MFMAs of tile j (worth 1.5 %: the vector
        // and the matrix instructions of the waves of a SIMD mostly take turns whatever their order -- SQ_VALU_MFMA_COEXEC_CYCLES 0.12 of the kernel's
        // cycles, MFMA-busy 0.38 + vector-busy 0.51 of them -- so what bounds this kernel is the SUM of its 30 MFMAs and ~310 vector instructions per
        // wave and chunk, profiles/r06_stem_wgrad_x6.txt)
        u4 B[2][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) { uint32_t h_, m_, l_; x6_split2(b[0][2 * q], b[0][2 * q + 1], h_, m_, l_); B[0][0][q] = h_; B[0][1][q] = m_; B[0][2][q] = l_; }
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            const int cur = j & 1;
            if (j + 1 < NACC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { uint32_t h_, m_, l_; x6_split2(b[j + 1][2 * q], b[j + 1][2 * q + 1], h_, m_, l_); B[cur ^ 1][0][q] = h_; B[cur ^ 1][1][q] = m_; B[cur ^ 1][2][q] = l_; }
            }
            WG_MF(A[0], B[cur][2], acc_lo[j]); WG_MF(A[2], B[cur][0], acc_lo[j]); WG_MF(A[1], B[cur][1], acc_lo[j]);
            WG_MF(A[0], B[cur][1], acc_lo[j]); WG_MF(A[1], B[cur][0], acc_lo[j]);
            WG_MF(A[0], B[cur][0], acc_hi[j]);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);       // ... eight vector instructions in its shadow
            }
        }
    };
#undef WG_MF
    gload(std::integral_constant<int, 0>{});
    if (c_begin + 1 < c_end) gload(std::integral_constant<int, 1>{});
    for (int c = c_begin; c < c_end; c += 3) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < c_end) chunk(std::integral_constant<int, 1>{}, c + 1);
        if (c + 2 < c_end) chunk(std::integral_constant<int, 2>{}, c + 2);
    }
    __syncthreads();                           // every wave is behind its last chunk
    const size_t tap0 = (size_t)((kt_ + p.wk0_t) * p.KH + p.wk0_h) * KW * 4;       // g offset of (kt, kh = 0, kw = 0, cs = 0)
    // the two sub-chunk waves hold partial sums of the SAME outputs: wave wk = 1 hands its sums to wave wk = 0 through LDS, one accumulator
    // per round
    float* image = p.g + (size_t)slice * p.wss;
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const f32x16 acc = acc_hi[j] + acc_lo[j];
        if (wk == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[wm][r][lane] = acc[r];
        }
        __syncthreads();
        if (wk == 0 && gcol[j] >= 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Cd) {
                    float* dst = image + (size_t)m * p.taps_full * 4 + tap0 + gcol[j];
                    const float v = acc[r] + xch[wm][r][lane];
                    if (p.wss) *dst = v;
                    else atomicAdd(dst, v);
                }
            }
        }
        __syncthreads();
    }
}

// Which kernel family pc_conv_wgrad gives a problem to -- ONE classification for pc_conv_wgrad, pc_conv_wgrad_multi and the host-side
// work accounting (pc_wgrad_work), so the A/B switches mean the same everywhere and the fp32 atomic sum order the goldens were
// made with cannot drift between copies of the predicates.
enum WgRoute { WG_STEM, WG_ROW3, WG_ROW9, WG_GENERIC };
#ifdef PICONS_DIAG
inline bool wg_ablate() { static const bool a = getenv("PICONS_WGRAD_ABLATE") != nullptr; return a; }
#else
constexpr bool wg_ablate() { return false; }
#endif
inline WgRoute wg_route(const pc_wgrad_desc* d) {
    static const int s4_env = getenv("PICONS_WGRAD_STEM") ? atoi(getenv("PICONS_WGRAD_STEM")) : 1;
    static const int row_env = getenv("PICONS_WGRAD_ROW") ? atoi(getenv("PICONS_WGRAD_ROW")) : 1;
    if (s4_env && d->Cs == 4 && d->lds % 4 == 0 && d->KW == 7 && d->ntap[2] == 7 && d->wk0[2] == 0 && d->istr[2] == 2 && d->ntap[1] == 7 &&
        d->ntap[0] <= 10 && d->istep[0] == 1 && d->istep[1] == 1 && d->istep[2] == 1 && d->Td == 0 && d->nbatch <= 1 && d->splitk >= 0 &&
        d->Wq % 28 == 0 && !wg_ablate())
        return WG_STEM;
    const bool csb64 = d->Cs % 64 == 0, csb32 = d->Cs % 32 == 0 && d->Cd > 64;
    const int padw = -d->ioff0[2], nprob = d->nbatch > 1 ? d->nbatch : 1;
    const bool row3 = d->KW == 3 && padw == 1 && d->Wq == d->Ws && (csb64 || csb32) && d->Ws % 28 == 0 && nprob == 1 && d->splitk >= 0;
    const bool row9 = d->KW == 9 && padw == 0 && d->Wq == d->Ws - 8 && d->Wq == 20 && csb64 && d->Tq == 1 && d->Hq == 1;   // spectral forms
    if (row_env && !wg_ablate() && d->Td == 0 && d->ntap[2] == d->KW && d->wk0[2] == 0 && d->istr[2] == 1 && d->istep[0] == 1 && d->istep[1] == 1 &&
        d->istep[2] == 1 && (row3 || row9))
        return row9 ? WG_ROW9 : WG_ROW3;
    return WG_GENERIC;
}
// the stem's weight gradient on the bf16 matrix cores (wgrad4_x6_kernel): asked for (PC_WG_X6) and the 4th source channel is padding (PC_WG_CS3:
// the kernel packs the 3 real channels of the 7 x 7 taps into five accumulators)
inline bool wg_stem_x6(const pc_wgrad_desc* d) {
    static const int pack3 = getenv("PICONS_WGRAD_STEM_PACK3") ? atoi(getenv("PICONS_WGRAD_STEM_PACK3")) : 1;
    static const int x6env = getenv("PICONS_WGRAD_STEM_X6") ? atoi(getenv("PICONS_WGRAD_STEM_X6")) : 1;
    return x6env && pack3 && (d->flags & PC_WG_X6) && (d->flags & PC_WG_CS3) && wg_route(d) == WG_STEM;
}
// does the problem's launch multiply on the bf16 matrix cores?  (the 9-tap spectral planes and un-flagged problems stay on fp32 MFMA)
inline bool wg_uses_x6(const pc_wgrad_desc* d) {
    if (!(d->flags & PC_WG_X6)) return false;
    const WgRoute r = wg_route(d);
    return r == WG_ROW3 || r == WG_GENERIC || (r == WG_STEM && wg_stem_x6(d));
}
// 256-column tiles with 16-position chunks for the long-K launches of the generic kernel (PICONS_WGRAD_WIDE bit 0: 128-row, bit 1: 64-row tiles)
inline bool wg_wide(const pc_wgrad_desc* d, bool small_m) {
    if (d->flags & PC_WG_X6) return false;          // the bf16-split kernel keeps 128-column tiles (two accumulators per tile)
    static const int wide_env = getenv("PICONS_WGRAD_WIDE") ? atoi(getenv("PICONS_WGRAD_WIDE")) : 3;
    const int64_t P = (int64_t)d->N * d->Tq * d->Hq * d->Wq;
    const int Ntot = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Cs;
    return !wg_ablate() && Ntot >= 512 && (((wide_env & 1) && !small_m) || ((wide_env & 2) && small_m)) && P >= 65536;
}
// row-segment kernel geometry (shared by the launch and the accounting)
struct Row3Geo { bool csb64, small_m, x6; int bkp, bm, csb; };
inline Row3Geo wg_row_geo(const pc_wgrad_desc* d, bool row9) {
    Row3Geo r;
    r.csb64 = d->Cs % 64 == 0;
    // 64-row tiles also when they pad at least 20 % less than 128-row tiles (192 = 3 x 64 vs 2 x 128)
    r.small_m = d->Cd <= 64 || row9 || (r.csb64 && cdiv(d->Cd, 64) * 64 * 5 <= cdiv(d->Cd, 128) * 128 * 4);
    r.bkp = row9 ? 20 : ((r.csb64 && r.small_m && d->Ws % 56 == 0) ? 56 : 28);
    r.bm = r.small_m ? 64 : 128;
    r.csb = r.csb64 ? 64 : 32;
    r.x6 = !row9 && (d->flags & PC_WG_X6) != 0;
    if (r.x6) {          // bf16-split kernel: one 32-channel tile of D and of S per wave, whole k16 steps per chunk (the segment's tail reads zeros)
        r.bkp = d->Ws % 56 == 0 ? 64 : 32;
        r.bm = r.csb64 ? 64 : 128;
        r.small_m = r.csb64;
    }
    return r;
}
// the generic kernel's launches for a problem: up to two row ranges [lo, hi) with 64- or 128-row tiles
struct WgSplit { int n; int lo[2], hi[2]; bool small_m[2]; };
inline WgSplit wg_generic_split(const pc_wgrad_desc* d) {
    // 128-row tiles for the bulk.  When the grid is many rounds deep without split-K (PrimaryCaps: 544 = 4*128 + 32
    // rows x 527 column tiles), a remainder of at most 64 channels gets its own launch with 64-row tiles instead
    // of a padded 128-row tile (4.98 vs 5.23 ms); small problems lose more to the second launch than they save
    const int Ntot = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Cs;
    const int full = d->Cd / 128 * 128, rem = d->Cd - full;
    const bool deep = (int64_t)(full / 128) * cdiv(Ntot, 128) * (d->nbatch > 1 ? d->nbatch : 1) >= 1024;
    // 64-row tiles run at ~0.88 of the 128-row tiles' rate: take them when they pad that much less (192 = 3 x 64 vs 2 x 128)
    const double pad128 = (double)cdiv(d->Cd, 128) * 128, pad64 = (double)cdiv(d->Cd, 64) * 64;
    WgSplit w; w.n = 1; w.lo[0] = 0; w.hi[0] = d->Cd; w.small_m[0] = false;
    if (d->Cd <= 64) w.small_m[0] = true;
    else if (rem != 0 && rem <= 64 && deep) { w.n = 2; w.hi[0] = full; w.lo[1] = full; w.hi[1] = d->Cd; w.small_m[1] = true; }
    else if (pad128 > 1.15 * pad64) w.small_m[0] = true;
    return w;
}

}  // namespace

// Host-only work accounting of one pc_conv_wgrad call (no GPU call), see pc_conv_work.  out[5]: multiply-accumulates ISSUED to
// the matrix cores (whole tiles, padded chunks; chunks the row-segment / stem kernels never multiply because their (kt, kh) tap
// reads outside the volume are not counted), EXECUTED on real rows x real columns of the same chunks, VALID (non-padding source
// positions only), the route (0 stem, 1 row-segment 3 taps, 2 row-segment 9 taps, 3 generic split-K) and the number of kernel launches.
extern "C" int pc_wgrad_work(const pc_wgrad_desc* d, int cd_real, int cs_real, double* out) {
    PC_CHECK_ARG(d && out, "pc_wgrad_work: null pointer");
    if (cd_real <= 0) cd_real = d->Cd;
    if (cs_real <= 0) cs_real = d->Cs;
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    const int Q[3] = {d->Tq, d->Hq, d->Wq}, I[3] = {d->Ts, d->Hs, d->Ws};
    // per dimension and tap: lattice points whose source position exists
    double V[3][16] = {}, Vsum[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        PC_CHECK_ARG(d->ntap[k] >= 1 && d->ntap[k] <= 16, "pc_wgrad_work: ntap out of range");
        for (int a = 0; a < d->ntap[k]; ++a) {
            int cnt = 0;
            for (int q = 0; q < Q[k]; ++q) cnt += (unsigned)(q * d->istr[k] + d->ioff0[k] + a * d->istep[k]) < (unsigned)I[k];
            V[k][a] = cnt; Vsum[k] += cnt;
        }
    }
    const double valid = (double)d->N * nb * Vsum[0] * Vsum[1] * Vsum[2] * cd_real * cs_real;
    const WgRoute route = wg_route(d);
    double issued = 0, executed = 0, launches = 1;
    if (route == WG_STEM) {
        const bool x6 = wg_stem_x6(d);
        const int nseg = x6 ? cdiv(d->Wq, 16) : d->Wq / 28, mt = cdiv(d->Cd, 64);
        const double seg = x6 ? 16.0 : 28.0;
        static const int pack3 = getenv("PICONS_WGRAD_STEM_PACK3") ? atoi(getenv("PICONS_WGRAD_STEM_PACK3")) : 1;
        const bool p3 = (d->flags & PC_WG_CS3) && pack3;
        const double nacc = p3 ? (7 * 7 * 3 + 31) / 32 : 7;
        for (int a = 0; a < d->ntap[0]; ++a) {
            const double chunks = (double)d->N * V[0][a] * d->Hq * nseg;          // every kh of the kt tap, all w
            issued += chunks * seg * (mt * 64.0) * nacc * 32.0;
            executed += (double)d->N * V[0][a] * d->Hq * d->Wq * cd_real * (7.0 * 7.0 * cs_real);
        }
    } else if (route == WG_ROW3 || route == WG_ROW9) {
        const Row3Geo r = wg_row_geo(d, route == WG_ROW9);
        const int nseg = cdiv(d->Wq, r.bkp), mt = cdiv(d->Cd, r.bm), ncs = d->Cs / r.csb;
        for (int a = 0; a < d->ntap[0]; ++a)
            for (int b = 0; b < d->ntap[1]; ++b) {
                const double chunks = (double)d->N * nb * V[0][a] * V[1][b] * nseg;
                issued += chunks * r.bkp * (mt * (double)r.bm) * ((double)d->KW * r.csb * ncs);
                executed += chunks * (r.x6 ? (double)d->Wq / nseg : (double)r.bkp) * cd_real * ((double)d->KW * cs_real);       // real positions only
            }
    } else {
        const int64_t P = (int64_t)d->N * d->Tq * d->Hq * d->Wq;
        const int Ntot = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Cs;
        const WgSplit w = wg_generic_split(d);
        launches = w.n;
        for (int i = 0; i < w.n; ++i) {
            const bool wide = wg_wide(d, w.small_m[i]);
            const int bm = w.small_m[i] ? 64 : 128, bn = wide ? 256 : 128, kb = wide ? 16 : BK;
            issued += (double)cdiv(P, kb) * kb * ((double)cdiv(w.hi[i] - w.lo[i], bm) * bm) * ((double)cdiv(Ntot, bn) * bn) * nb;
        }
        executed = (double)P * nb * cd_real * ((double)d->ntap[0] * d->ntap[1] * d->ntap[2] * cs_real);
    }
    out[0] = issued; out[1] = executed; out[2] = valid; out[3] = (double)route; out[4] = launches;
    return PC_OK;
}

namespace {
// floats of one K-slice image of a problem's gradient: g's own layout [Cd][KT*KH*KW][Cs] (x the problems of a batched launch)
inline long long wg_image_floats(const pc_wgrad_desc* d) {
    return d->nbatch > 1 ? (long long)d->gbstride * d->nbatch : (long long)d->Cd * d->KT * d->KH * d->KW * d->Cs;
}
}  // namespace

// weight-gradient launch: carries pc_run_ops_timed's event pair in its own dispatch when one is set (as the conv launches do)
#define WG_LAUNCH(kernel, grid, s, arg)                                                                        \
    do {                                                                                                       \
        if (pc_tl_ev_start) hipExtLaunchKernelGGL(kernel, grid, dim3(256), 0, s, pc_tl_ev_start, pc_tl_ev_stop, 0, arg); \
        else hipLaunchKernelGGL(kernel, grid, dim3(256), 0, s, arg);                                           \
    } while (0)

// dry != NULL: no launch -- *dry = the number of K slices (workspace images) the launch would write (pc_wgrad_slices)
static int wgrad_run(const pc_wgrad_desc* d, const float* D, const float* S, float* g, pc_stream s_, int* dry) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(d && (dry || (D && S && g)), "pc_conv_wgrad: null pointer");
    PC_CHECK_ARG(d->Cd % 4 == 0 && d->Cs % 4 == 0 && d->ldd % 4 == 0 && d->lds % 4 == 0,
                 "pc_conv_wgrad: channel counts / strides must be multiples of 4 (Cd=%d Cs=%d)", d->Cd, d->Cs);
    PC_CHECK_ARG(dry || (((uintptr_t)D % 16 == 0) && ((uintptr_t)S % 16 == 0)), "pc_conv_wgrad: D/S must be 16-byte aligned");
    PC_CHECK_ARG(d->ws_slices >= 0 && !(d->ws_slices > 0 && d->splitk == -1), "pc_conv_wgrad: ws_slices = %d with splitk = %d", d->ws_slices, d->splitk);
    // ws_slices > 0: g is a workspace of that many K-slice images (wg_image_floats apart); slice k writes image k with plain stores
    const long long wss = d->ws_slices > 0 ? wg_image_floats(d) : 0;
    int nslices = 1;
    WgK k;
    k.D = D; k.S = S; k.g = g;
    k.N = d->N; k.Tq = d->Tq; k.Hq = d->Hq; k.Wq = d->Wq; k.Cd = d->Cd; k.ldd = d->ldd;
    k.Ts = d->Ts; k.Hs = d->Hs; k.Ws = d->Ws; k.Cs = d->Cs; k.lds = d->lds;
    for (int i = 0; i < 3; ++i) { k.istr[i] = d->istr[i]; k.ntap[i] = d->ntap[i]; k.ioff0[i] = d->ioff0[i]; k.istep[i] = d->istep[i]; k.wk0[i] = d->wk0[i]; }
    PC_CHECK_ARG(d->nbatch <= 4096, "pc_conv_wgrad: nbatch too large");
    PC_CHECK_ARG(d->splitk != -1 || (d->ntap[0] == d->KT && d->ntap[1] == d->KH && d->ntap[2] == d->KW), "pc_conv_wgrad: splitk = -1 (plain stores) needs every tap present");
    PC_CHECK_ARG(d->wk0[0] + d->ntap[0] <= d->KT && d->wk0[1] + d->ntap[1] <= d->KH && d->wk0[2] + d->ntap[2] <= d->KW, "pc_conv_wgrad: trimmed taps exceed the weight extents");
    k.KH = d->KH; k.KW = d->KW; k.NtotFull = d->KT * d->KH * d->KW * d->Cs;
    k.dlat = d->Td > 0; k.Td = d->Td; k.Hd = d->Hd; k.Wd = d->Wd;
    for (int i = 0; i < 3; ++i) k.doff[i] = d->doff[i];
    PC_CHECK_ARG(!k.dlat || (int64_t)d->N * d->Td * d->Hd * d->Wd < (1ll << 31), "pc_conv_wgrad: D tensor too large");
    const int64_t P = (int64_t)d->N * d->Tq * d->Hq * d->Wq;
    PC_CHECK_ARG(P > 0 && P < (1ll << 31) && (int64_t)d->N * d->Ts * d->Hs * d->Ws < (1ll << 31), "pc_conv_wgrad: position count out of range");
    PC_CHECK_ARG((int64_t)d->N * d->Ts * d->Hs * d->Ws * d->lds * 4 < DMA_MAX_BYTES && (k.dlat ? (int64_t)d->N * d->Td * d->Hd * d->Wd : P) * d->ldd * 4 < DMA_MAX_BYTES,
                 "pc_conv_wgrad: D or S exceeds the 4 GiB the LDS-DMA gather addresses: use a smaller per-GPU batch");
    k.P = (int)P;
    k.Ntot = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Cs;
    k.nchunks = cdiv(P, BK);
#ifdef PICONS_DIAG
    static const int abl = getenv("PICONS_WGRAD_ABLATE") ? atoi(getenv("PICONS_WGRAD_ABLATE")) : 0;   // diagnostic build: no tile fetch in the K loop (wrong results)
#endif
    // rows of D's channels [m_lo, m_hi) with 64- or 128-row tiles
    // conv with 3 taps, stride 1, padding 1 along w and 32- / 64-channel blocks of S: row-segment kernel (the kw taps share
    // one LDS tile)
    // split-K fills `rounds` rounds of resident-block slots: one round measured best on the step (fewer atomics per weight;
    // the second lane's launches fill what a single round leaves idle)
    static const double rounds = getenv("PICONS_WGRAD_ROUNDS") ? atof(getenv("PICONS_WGRAD_ROUNDS")) : 1.0;
    const WgRoute route = wg_route(d);
    if (route == WG_STEM) {
        Wg4K q;
        q.D = D; q.S = S; q.g = g;
        q.N = d->N; q.T = d->Tq; q.H = d->Hq; q.W = d->Wq; q.Cd = d->Cd; q.ldd = d->ldd; q.lds = d->lds;
        q.ntap_t = d->ntap[0]; q.wk0_t = d->wk0[0]; q.wk0_h = d->wk0[1]; q.KH = d->KH;
        q.Ts = d->Ts; q.Hs = d->Hs; q.Wsw = d->Ws; q.istr_t = d->istr[0]; q.istr_h = d->istr[1]; q.ioff_t = d->ioff0[0]; q.ioff_h = d->ioff0[1];
        q.padw = -d->ioff0[2];
        static const int pack3 = getenv("PICONS_WGRAD_STEM_PACK3") ? atoi(getenv("PICONS_WGRAD_STEM_PACK3")) : 1;
        const bool x6 = wg_stem_x6(d);                  // bf16-split kernel: chunks of two 16-position sub-chunks
        q.nseg = x6 ? cdiv(d->Wq, 16) : d->Wq / 28; q.mt = cdiv(d->Cd, 64); q.taps_full = d->KT * d->KH * d->KW;
        // K slices per kt in proportion to the output t planes whose source plane exists
        int ntv[10], tot = 0;
        for (int a = 0; a < q.ntap_t; ++a) {
            int cnt = 0;
            for (int t = 0; t < d->Tq; ++t) { const int ts = t * q.istr_t + q.ioff_t + a; cnt += ts >= 0 && ts < d->Ts; }
            ntv[a] = cnt; tot += cnt;
        }
        PC_CHECK_ARG(tot > 0, "pc_conv_wgrad: no valid tap");
        static const int s4_slots = getenv("PICONS_WGRAD_STEM_SLOTS") ? atoi(getenv("PICONS_WGRAD_STEM_SLOTS")) : 768;   // 162 VGPRs, 28 KiB LDS: 3 blocks per CU
        const int slots = d->splitk > 0 ? d->splitk * q.ntap_t : (x6 ? 512 : s4_slots) / q.mt;                         // the bf16-split kernel: two blocks per CU
        q.pre[0] = 0;
        for (int a = 0; a < q.ntap_t; ++a) {
            int sl = ntv[a] ? (int)((double)slots * ntv[a] / tot) : 0;
            const int64_t nch = x6 ? ((int64_t)d->N * ntv[a] * d->Hq * q.nseg + 1) / 2 : (int64_t)d->N * ntv[a] * d->Hq * q.nseg;
            if (ntv[a] && sl < 1) sl = 1;
            if (sl > nch / 4 && nch >= 4) sl = (int)(nch / 4);
            if (ntv[a] && sl < 1) sl = 1;
            q.pre[a + 1] = q.pre[a] + sl;
        }
        const dim3 grid((unsigned)(q.pre[q.ntap_t] * q.mt));
        for (int a = 0; a < q.ntap_t; ++a) nslices = std::max(nslices, q.pre[a + 1] - q.pre[a]);
        if (dry) { *dry = nslices; return PC_OK; }
        PC_CHECK_ARG(!wss || nslices <= d->ws_slices, "pc_conv_wgrad: the workspace holds %d slice images, this launch writes %d (pc_wgrad_slices)", d->ws_slices, nslices);
        q.wss = wss;
        {   // block order: slices sorted by where in the volume they are (their relative position inside their kt's slice range), kt as tie-break
            static const int il = getenv("PICONS_WGRAD_STEM_INTERLEAVE") ? atoi(getenv("PICONS_WGRAD_STEM_INTERLEAVE")) : 1;
            const int tot_sl = q.pre[q.ntap_t];
            q.interleave = il && tot_sl <= 1024;
            if (q.interleave) {
                std::pair<double, int> key[1024];
                for (int a = 0; a < q.ntap_t; ++a) {
                    const int ns = q.pre[a + 1] - q.pre[a];
                    for (int i = 0; i < ns; ++i) key[q.pre[a] + i] = std::make_pair((i + 0.5) / ns, q.pre[a] + i);
                }
                std::sort(key, key + tot_sl);
                for (int i = 0; i < tot_sl; ++i) q.order[i] = (unsigned short)key[i].second;
            }
        }
        if (x6) WG_LAUNCH((wgrad4_x6_kernel<7, 2, 7>), grid, s, q);
        else if ((d->flags & PC_WG_CS3) && pack3) WG_LAUNCH((wgrad4_kernel<28, 7, 2, 7, true>), grid, s, q);
        else WG_LAUNCH((wgrad4_kernel<28, 7, 2, 7, false>), grid, s, q);
        PC_CHECK_LAUNCH("wgrad4_kernel");
        return PC_OK;
    }
    if (route == WG_ROW3 || route == WG_ROW9) {
        const bool row9 = route == WG_ROW9;
        const int padw = -d->ioff0[2], nprob = d->nbatch > 1 ? d->nbatch : 1;
        Wg3K q;
        q.D = D; q.S = S; q.g = g;
        q.N = d->N; q.T = d->Tq; q.H = d->Hq; q.W = d->Wq; q.Cd = d->Cd; q.ldd = d->ldd; q.Cs = d->Cs; q.lds = d->lds;
        q.Ts = d->Ts; q.Hs = d->Hs; q.istr_t = d->istr[0]; q.istr_h = d->istr[1]; q.ioff_t = d->ioff0[0]; q.ioff_h = d->ioff0[1];
        q.ntap_t = d->ntap[0]; q.ntap_h = d->ntap[1]; q.wk0_t = d->wk0[0]; q.wk0_h = d->wk0[1]; q.KH = d->KH;
        q.taps_full = d->KT * d->KH * d->KW;
        q.Wsw = d->Ws; q.padw = padw; q.nprob = nprob; q.dbs = d->dbstride; q.sbs = d->sbstride; q.gbs = d->gbstride;
        const Row3Geo geo = wg_row_geo(d, row9);
        const bool csb64 = geo.csb64, small_m = geo.small_m;
        const int bkp = geo.bkp;
        q.nseg = cdiv(d->Wq, bkp);
        q.nchunks = d->N * d->Tq * d->Hq * q.nseg;
        q.mt = cdiv(d->Cd, geo.bm); q.ncs = d->Cs / (csb64 ? 64 : 32);
        const int64_t tiles = (int64_t)q.mt * q.ncs * q.ntap_t * q.ntap_h;
        // resident-block slots of the variant that will run: the fp32 kernels hold three blocks per CU (two with 56-position segments), the
        // bf16-split ones two -- except wgrad3_x6_kernel<128, 64, 32, 4>, whose 82 KiB of LDS admit ONE block per CU
        const int slots = geo.x6 ? ((!csb64 && bkp == 64) ? 256 : 512) : (bkp == 56 ? 512 : 768);
        int splitk = d->splitk > 0 ? d->splitk : (d->splitk == -1 ? 1 : (int)(rounds * slots / (tiles * nprob)));
        const int maxsplit = q.nchunks / 8 > 0 ? q.nchunks / 8 : 1;
        if (splitk > maxsplit) splitk = maxsplit;
        if (splitk < 1) splitk = 1;
        q.chunks_per_split = cdiv(q.nchunks, splitk);
        q.nsplit = cdiv(q.nchunks, q.chunks_per_split);
        q.store = d->splitk == -1;
        if (dry) { *dry = q.nsplit; return PC_OK; }
        PC_CHECK_ARG(!wss || q.nsplit <= d->ws_slices, "pc_conv_wgrad: the workspace holds %d slice images, this launch writes %d (pc_wgrad_slices)", d->ws_slices, q.nsplit);
        q.wss = wss;
        const dim3 grid((unsigned)(tiles * q.nsplit * nprob));
        if (geo.x6) {
            if (csb64 && bkp == 64) WG_LAUNCH((wgrad3_x6_kernel<64, 64, 64, 2>), grid, s, q);
            else if (csb64) WG_LAUNCH((wgrad3_x6_kernel<64, 32, 64, 2>), grid, s, q);
            else if (bkp == 64) WG_LAUNCH((wgrad3_x6_kernel<128, 64, 32, 4>), grid, s, q);
            else WG_LAUNCH((wgrad3_x6_kernel<128, 32, 32, 4>), grid, s, q);
            PC_CHECK_LAUNCH("wgrad3_x6_kernel");
            return PC_OK;
        }
        if (row9) WG_LAUNCH((wgrad3_kernel<64, 20, 64, 2, 9>), grid, s, q);
        else if (!csb64) WG_LAUNCH((wgrad3_kernel<128, 28, 32, 4>), grid, s, q);
        else if (small_m && bkp == 56) WG_LAUNCH((wgrad3_kernel<64, 56, 64, 2>), grid, s, q);
        else if (small_m) WG_LAUNCH((wgrad3_kernel<64, 28, 64, 2>), grid, s, q);
        else WG_LAUNCH((wgrad3_kernel<128, 28, 64, 2>), grid, s, q);
        PC_CHECK_LAUNCH("wgrad3_kernel");
        return PC_OK;
    }
    bool ws_short = false;
    auto launch = [&](int m_lo, int m_hi, bool small_m) {
        // 256-column tiles with 16-position chunks for the long-K launches: 17-25 % less tile traffic per FLOP.  The kernel
        // is bound by the LDS-DMA fill rate, not by the MFMA loop (fetch ablation: 148 TF/s without the fetch, 113-118 with either
        // operand tile alone, 103 with both): 3x3x3 128->128 @112^2 103 -> 115 TF/s, the 64-channel layers 87 -> 92
        const bool wide = wg_wide(d, small_m);
        const int bm = small_m ? 64 : 128, bn = wide ? 256 : 128;
        k.nchunks = cdiv(P, wide ? 16 : BK);
        const int mt = cdiv(m_hi - m_lo, bm), ntl = cdiv(k.Ntot, bn);
        int splitk = d->splitk;
        const int nb = d->nbatch > 1 ? d->nbatch : 1;
        if (splitk == -1) splitk = 1;
        else if (splitk <= 0) {
            // resident blocks per CU follow the LDS footprint (64 KiB -> 2, 48 KiB -> 3): fill whole rounds of
            // slots and never spill a few blocks into the next (1026 blocks ran ~25 % slower than 1022); at least
            // 8 chunks (256 positions) per slice
            const int64_t tiles = (int64_t)mt * ntl * nb;
            const int slots = (small_m || wide) ? 768 : 512;
            splitk = (int)(rounds * slots / tiles);
            const int maxsplit = k.nchunks / 8 > 0 ? k.nchunks / 8 : 1;
            if (splitk > maxsplit) splitk = maxsplit;
            if (splitk < 1) splitk = 1;
        }
        WgK kk = k;
        kk.mbase = m_lo; kk.mend = m_hi;
        kk.store = d->splitk == -1;
        kk.dbs = nb > 1 ? d->dbstride : 0; kk.sbs = nb > 1 ? d->sbstride : 0; kk.gbs = nb > 1 ? d->gbstride : 0;
        kk.chunks_per_split = cdiv(k.nchunks, splitk);
        splitk = cdiv(k.nchunks, kk.chunks_per_split);
        kk.nsplit = splitk;
        kk.mt = mt; kk.ntl = ntl;
        nslices = std::max(nslices, splitk);
        if (dry) return;
        if (wss && splitk > d->ws_slices) { ws_short = true; return; }
        kk.wss = wss;
        dim3 grid((unsigned)((int64_t)mt * ntl * nb * splitk));
#ifdef PICONS_DIAG
        if (abl == 2) {          // + fragment reads hoisted out of the k-step loop
            if (small_m) WG_LAUNCH((wgrad_kernel<64, 128, 2>), grid, s, kk);
            else WG_LAUNCH((wgrad_kernel<128, 128, 2>), grid, s, kk);
        } else if (abl >= 4 && abl <= 6) {
            if (abl == 4) { if (small_m) WG_LAUNCH((wgrad_kernel<64, 128, 4>), grid, s, kk); else WG_LAUNCH((wgrad_kernel<128, 128, 4>), grid, s, kk); }
            if (abl == 5) { if (small_m) WG_LAUNCH((wgrad_kernel<64, 128, 5>), grid, s, kk); else WG_LAUNCH((wgrad_kernel<128, 128, 5>), grid, s, kk); }
            if (abl == 6) { if (small_m) WG_LAUNCH((wgrad_kernel<64, 128, 6>), grid, s, kk); else WG_LAUNCH((wgrad_kernel<128, 128, 6>), grid, s, kk); }
        } else if (abl == 3) {   // + no barrier per chunk
            if (small_m) WG_LAUNCH((wgrad_kernel<64, 128, 3>), grid, s, kk);
            else WG_LAUNCH((wgrad_kernel<128, 128, 3>), grid, s, kk);
        } else if (abl) {
            if (small_m) WG_LAUNCH((wgrad_kernel<64, 128, 1>), grid, s, kk);
            else WG_LAUNCH((wgrad_kernel<128, 128, 1>), grid, s, kk);
        } else
#endif
        if (d->flags & PC_WG_X6) {
#ifdef PICONS_DIAG
            // diagnostic library only (tools/wgrad_x6_acc_probe.py): flag bit 4 = one accumulator per tile instead of the hi / lo pair -- 1.1x the
            // fp32 kernel's distance from fp64 where the pair is at 0.5 - 0.6x, at the same speed
            if (d->flags & 4) {
                if (small_m) WG_LAUNCH((wgrad_x6_kernel<64, 128, false>), grid, s, kk);
                else WG_LAUNCH((wgrad_x6_kernel<128, 128, false>), grid, s, kk);
            } else
#endif
            if (small_m) WG_LAUNCH((wgrad_x6_kernel<64, 128>), grid, s, kk);
            else WG_LAUNCH((wgrad_x6_kernel<128, 128>), grid, s, kk);
        } else if (small_m && wide) WG_LAUNCH((wgrad_kernel<64, 256, 0, 16>), grid, s, kk);
        else if (small_m) WG_LAUNCH((wgrad_kernel<64, 128>), grid, s, kk);
        else if (wide) WG_LAUNCH((wgrad_kernel<128, 256, 0, 16>), grid, s, kk);
        else WG_LAUNCH((wgrad_kernel<128, 128>), grid, s, kk);
    };
    const WgSplit w = wg_generic_split(d);
    for (int i = 0; i < w.n; ++i) launch(w.lo[i], w.hi[i], w.small_m[i]);
    if (dry) { *dry = nslices; return PC_OK; }
    PC_CHECK_ARG(!ws_short, "pc_conv_wgrad: the workspace holds %d slice images, this launch writes %d (pc_wgrad_slices)", d->ws_slices, nslices);
    PC_CHECK_LAUNCH("wgrad_kernel");
    return PC_OK;
}

extern "C" int pc_conv_wgrad(const pc_wgrad_desc* d, const float* D, const float* S, float* g, pc_stream s) {
    return wgrad_run(d, D, S, g, s, nullptr);
}

extern "C" int pc_wgrad_uses_x6(const pc_wgrad_desc* d) { return d && wg_uses_x6(d) ? 1 : 0; }

extern "C" int pc_wgrad_slices(const pc_wgrad_desc* d) {
    int n = 0;
    pc_wgrad_desc e;
    if (!d) { pc_set_error("pc_wgrad_slices: null descriptor"); return -1; }
    e = *d; e.ws_slices = 0;
    return wgrad_run(&e, nullptr, nullptr, nullptr, nullptr, &n) == PC_OK ? n : -1;
}


// ---- several weight-gradient problems in one launch ------------------------------------------------------------------------
namespace {

// the problems pc_conv_wgrad would give to the generic split-K kernel (not the stem / row-segment kernels, not a forced split)
bool wg_is_generic(const pc_wgrad_desc* d) {
    if (wg_ablate() || wg_route(d) != WG_GENERIC || d->splitk != 0) return false;
    const WgSplit w = wg_generic_split(d);
    for (int i = 0; i < w.n; ++i)
        if (wg_wide(d, w.small_m[i])) return false;          // long-K launches keep their 256-column tiles
    return true;
}

int wg_fill(const pc_wgrad_desc* d, const float* D, const float* S, float* g, WgK& k) {
    PC_CHECK_ARG(d && D && S && g, "pc_conv_wgrad_multi: null pointer");
    PC_CHECK_ARG(d->Cd >= 1 && d->Cs >= 4 && d->Cs % 4 == 0 && d->ldd % 4 == 0 && d->lds % 4 == 0, "pc_conv_wgrad_multi: channel counts / strides must be multiples of 4");
    PC_CHECK_ARG(((uintptr_t)D % 16 == 0) && ((uintptr_t)S % 16 == 0), "pc_conv_wgrad_multi: D/S must be 16-byte aligned");
    PC_CHECK_ARG(d->wk0[0] + d->ntap[0] <= d->KT && d->wk0[1] + d->ntap[1] <= d->KH && d->wk0[2] + d->ntap[2] <= d->KW, "pc_conv_wgrad_multi: trimmed taps exceed the weight extents");
    k.D = D; k.S = S; k.g = g;
    k.N = d->N; k.Tq = d->Tq; k.Hq = d->Hq; k.Wq = d->Wq; k.Cd = d->Cd; k.ldd = d->ldd;
    k.Ts = d->Ts; k.Hs = d->Hs; k.Ws = d->Ws; k.Cs = d->Cs; k.lds = d->lds;
    for (int i = 0; i < 3; ++i) { k.istr[i] = d->istr[i]; k.ntap[i] = d->ntap[i]; k.ioff0[i] = d->ioff0[i]; k.istep[i] = d->istep[i]; k.wk0[i] = d->wk0[i]; k.doff[i] = d->doff[i]; }
    k.KH = d->KH; k.KW = d->KW; k.NtotFull = d->KT * d->KH * d->KW * d->Cs;
    k.dlat = d->Td > 0; k.Td = d->Td; k.Hd = d->Hd; k.Wd = d->Wd;
    const int64_t P = (int64_t)d->N * d->Tq * d->Hq * d->Wq;
    PC_CHECK_ARG(P > 0 && P < (1ll << 31) && (int64_t)d->N * d->Ts * d->Hs * d->Ws < (1ll << 31), "pc_conv_wgrad_multi: position count out of range");
    PC_CHECK_ARG((int64_t)d->N * d->Ts * d->Hs * d->Ws * d->lds * 4 < DMA_MAX_BYTES && (d->Td > 0 ? (int64_t)d->N * d->Td * d->Hd * d->Wd : P) * d->ldd * 4 < DMA_MAX_BYTES,
                 "pc_conv_wgrad_multi: D or S exceeds the 4 GiB the LDS-DMA gather addresses");
    PC_CHECK_ARG(!k.dlat || (int64_t)d->N * d->Td * d->Hd * d->Wd < (1ll << 31), "pc_conv_wgrad_multi: D tensor too large");
    k.P = (int)P;
    k.Ntot = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Cs;
    k.nchunks = cdiv(P, 32);
    k.mbase = 0; k.mend = d->Cd; k.store = 0; k.wss = 0;
    PC_CHECK_ARG(d->ws_slices == 0, "pc_conv_wgrad_multi: K-slice workspaces are not supported in grouped launches");
    const int nb = d->nbatch > 1 ? d->nbatch : 1;
    k.dbs = nb > 1 ? d->dbstride : 0; k.sbs = nb > 1 ? d->sbstride : 0; k.gbs = nb > 1 ? d->gbstride : 0;
    return PC_OK;
}

}  // namespace

extern "C" int pc_conv_wgrad_multi(const pc_wgrad_job* jobs, int njobs, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(jobs && njobs >= 1 && njobs <= 64, "pc_conv_wgrad_multi: jobs=%p njobs=%d (1..64)", (const void*)jobs, njobs);
    static const int multi_env = getenv("PICONS_WGRAD_MULTI") ? atoi(getenv("PICONS_WGRAD_MULTI")) : 1;
    std::vector<int> gen;
    for (int j = 0; j < njobs; ++j) {
        if (multi_env && wg_is_generic(&jobs[j].d)) { gen.push_back(j); continue; }
        const int rc = pc_conv_wgrad(&jobs[j].d, jobs[j].D, jobs[j].S, jobs[j].g, s_);      // stem / row-segment / long-K problems: own launch
        if (rc != PC_OK) return rc;
    }
    for (size_t g0 = 0; g0 < gen.size(); g0 += WG_MAXJOBS) {
        const int n = (int)std::min<size_t>(WG_MAXJOBS, gen.size() - g0);
        if (n == 1) {
            const pc_wgrad_job& q = jobs[gen[g0]];
            const int rc = pc_conv_wgrad(&q.d, q.D, q.S, q.g, s_);
            if (rc != PC_OK) return rc;
            continue;
        }
        WgMulti m;
        m.njobs = n;
        // 64- or 128-row tiles for the whole group: whichever pads less (64-row tiles run at ~0.88 of the 128-row rate)
        double work[2] = {0, 0};
        for (int j = 0; j < n; ++j) {
            const pc_wgrad_desc& d = jobs[gen[g0 + j]].d;
            const double cols = (double)cdiv(d.ntap[0] * d.ntap[1] * d.ntap[2] * d.Cs, 128) * 128, P = (double)d.N * d.Tq * d.Hq * d.Wq * (d.nbatch > 1 ? d.nbatch : 1);
            work[0] += (double)cdiv(d.Cd, 64) * 64 * cols * P;
            work[1] += (double)cdiv(d.Cd, 128) * 128 * cols * P;
        }
        const bool small_m = work[0] < 0.88 * work[1];
        const int bm = small_m ? 64 : 128, slots = small_m ? 768 : 512;
        const double total = work[small_m ? 0 : 1];
        int first = 0;
        for (int j = 0; j < n; ++j) {
            const pc_wgrad_job& q = jobs[gen[g0 + j]];
            WgK& k = m.job[j];
            const int rc = wg_fill(&q.d, q.D, q.S, q.g, k);
            if (rc != PC_OK) return rc;
            const int nb = q.d.nbatch > 1 ? q.d.nbatch : 1;
            k.mt = cdiv(q.d.Cd, bm); k.ntl = cdiv(k.Ntot, 128);
            const int64_t tiles = (int64_t)k.mt * k.ntl * nb;
            // blocks in proportion to the problem's (padded) work: every block of the grid gets about the same number of chunks
            const double w = (double)k.mt * bm * k.ntl * 128 * (double)k.P * nb;
            int splitk = (int)(slots * (w / total) / (double)tiles + 0.5);
            const int maxsplit = k.nchunks / 8 > 0 ? k.nchunks / 8 : 1;
            if (splitk > maxsplit) splitk = maxsplit;
            if (splitk < 1) splitk = 1;
            k.chunks_per_split = cdiv(k.nchunks, splitk);
            k.nsplit = cdiv(k.nchunks, k.chunks_per_split);
            m.first[j] = first;
            first += (int)(tiles * k.nsplit);
        }
        for (int j = n; j <= WG_MAXJOBS; ++j) m.first[j] = first;
        if (small_m) hipLaunchKernelGGL((wgrad_multi_kernel<64, 128>), dim3((unsigned)first), dim3(256), 0, s, m);
        else hipLaunchKernelGGL((wgrad_multi_kernel<128, 128>), dim3((unsigned)first), dim3(256), 0, s, m);
        PC_CHECK_LAUNCH("wgrad_multi_kernel");
    }
    return PC_OK;
}
