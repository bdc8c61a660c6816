// Gather-GEMM convolution with fp32 operands multiplied on the bf16 matrix cores (gfx950): every fp32 value is the exact sum of three bf16
// numbers h + m + l (8 + 8 + 8 significant bits), and a product a * b is accumulated in fp32 from the six bf16 products of weight
// >= 2^-16 (hh, hm, mh, hl, lh, mm) on v_mfma_f32_32x32x16_bf16.  Against an fp64 reference the result is closer than an fp32 FMA chain's
// (tools/x6_tile_probe.hip: 2.5e-8 vs 2.9e-8 of sum |a b|; tests/test_x6_gpu.py holds every converted shape to that bar), and it runs on a
// pipe that executes beside the vector ALU, where v_mfma_f32_32x32x2_f32 shares the fp32 vector lanes.
//
// Same descriptor, tap walk, tap box, LDS-DMA gather and epilogues as conv_gemm_glds_kernel (conv.hip).  What differs:
//   * B (weights) arrives PRE-SPLIT: three bf16 planes in the weight layout [Co][taps][ldw] (pc_split_planes), fetched by LDS-DMA into
//     [plane][BN][32 bf16] tiles whose 16-byte slots are XOR-swizzled by (row >> 2) & 3 -- a fragment is three ds_read_b128, no VALU;
//   * A (activations) stays fp32 in HBM and in LDS and is split in registers by the wave that owns the rows (WN = 1: every A element is split
//     once per block): 5.5 vector instructions per element (and / sub / and / sub / perm on pairs), 1.8 per MFMA;
//   * the K loop is a three-stage software pipeline inside every wave -- read the fragments of k16 step t + 1, split them, multiply step t --
//     with one MFMA and its share of the side work per slot, pinned (sched_barrier binds only the machine scheduler: empty volatile asm on
//     the inputs and outputs of a split unit keeps the IR passes from hoisting or sinking the pure ALU work);
//   * LDS reads run one step ahead of the MFMAs, so the two-buffer ring is as deep as a three-buffer one: the barrier sits at the head of
//     the odd phases (chunk c + 1 landed, every wave has read chunk c), behind it goes the DMA of chunk c + 2 -- BOTH operands since round 5
//     (the weight planes used to follow half a chunk later, which left them half a chunk to land: the waves of these kernels spent
//     37 - 43 % of their cycles parked at that barrier) -- into chunk c's buffers.
//   * TAIL SPLIT.  A launch of T tiles on S resident-block slots runs floor(T / S) full rounds and then a last round with T mod S blocks: the
//     320-row spectral planes in 128 x 64 tiles are 1107 = 2 x 512 + 83 blocks of 234 K chunks each -- the chip is 16 % full for a third of the
//     kernel.  With a workspace the last T mod S tiles (or all tiles of a launch that does not fill one round) are multiplied by `ksplit` blocks each,
//     one K slice per block; a block leaves its partial accumulators in the workspace, and the block that finds itself last at the tile's
//     counter adds all slices IN SLICE ORDER (deterministic, whichever block that is) and runs the ordinary epilogue -- so every epilogue
//     (ReLU, BatchNorm partials, channel-major stores) works unchanged.  The split blocks carry the highest hardware block ids: they are
//     dispatched last and fill the last round evenly.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>
#include <mutex>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) { x6_split2(x0, x1, h, m, l); }

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long long n, long long pstride) {
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        const f32x4 v = *(const f32x4*)(src + i);
        uint32_t h0, m0, l0, h1, m1, l1;
        split2(v[0], v[1], h0, m0, l0);
        split2(v[2], v[3], h1, m1, l1);
        *(uint2*)(dst + i) = make_uint2(h0, h1);
        *(uint2*)(dst + pstride + i) = make_uint2(m0, m1);
        *(uint2*)(dst + 2 * pstride + i) = make_uint2(l0, l1);
    }
}

// Many split jobs in one launch (the per-step weight planes of every layer): the jobs travel by value in the kernel arguments
constexpr int SP_JOBS = 48;
struct SplitJobK { const float* src; uint16_t* dst; long long n, pstride; };
struct SplitPack { SplitJobK j[SP_JOBS]; int first[SP_JOBS + 1]; int n; };
__global__ __launch_bounds__(256) void split_planes_multi_kernel(const SplitPack pk) {
    int k = 0;
    while (k + 1 < pk.n && (int)blockIdx.x >= pk.first[k + 1]) ++k;
    const SplitJobK& J = pk.j[k];
    const long long i = ((long long)(blockIdx.x - pk.first[k]) * 256 + threadIdx.x) * 4;
    if (i >= J.n) return;
    const f32x4 v = *(const f32x4*)(J.src + i);
    uint32_t h0, m0, l0, h1, m1, l1;
    split2(v[0], v[1], h0, m0, l0);
    split2(v[2], v[3], h1, m1, l1);
    *(uint2*)(J.dst + i) = make_uint2(h0, h1);
    *(uint2*)(J.dst + J.pstride + i) = make_uint2(m0, m1);
    *(uint2*)(J.dst + 2 * J.pstride + i) = make_uint2(l0, l1);
}

__device__ __forceinline__ dma_rsrc_t dma_rsrc_h(const uint16_t* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffff00u, 0x00020000);
}
__device__ __forceinline__ void glds16h(dma_rsrc_t rs, unsigned voff, uint8_t* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}

// Tail split (below): tiles [0, full) are one block each; tile full + i, i < rem, is multiplied by `ksplit` blocks (one K slice each) that leave
// their partial accumulators in ws and count up ctr[i]; the block that arrives last adds the slices in slice order and runs the epilogue.
struct ConvX6 { ConvK k; const uint16_t* wp; long long wpstride; int full, rem, ksplit; float* ws; unsigned* ctr; int mfast; };

#define PC_MFX6(X, Y, Cc) Cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, X), __builtin_bit_cast(bf16x8, Y), Cc, 0, 0, 0)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_x6_kernel(const ConvX6 px) {
    const ConvK& p = px.k;
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NM = 6 * TM * TN, NU = 4 * TM;
    static_assert(TM >= 1 && TN >= 1 && BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiles");
    constexpr int APIECES = BM / 8, BPIECES = 3 * BN / 16;                 // 1 KiB DMA pieces per chunk
    constexpr int AR = (APIECES + NW - 1) / NW, BR = (BPIECES + NW - 1) / NW;
    static_assert(APIECES % NW == 0, "A pieces per wave");
    constexpr int ABYTES = BM * BK * 4, BPLANE = BN * BK * 2, BBYTES = 3 * BPLANE;
    constexpr int OPBYTES = 2 * (ABYTES + BBYTES);
    // output staging: BM / EH rows at a time
    constexpr int EH = (BM * BN * 4 <= OPBYTES) ? 1 : 2;
    static_assert(BM * BN * 4 / EH <= OPBYTES, "output tile fits the operand buffers");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
    uint8_t* As0 = smem8;                              // [2][BM][32] fp32
    uint8_t* Bs0 = smem8 + 2 * ABYTES;                 // [2][3][BN][32] bf16
    int* rinfo = (int*)(smem8 + OPBYTES);              // [BM][4] n,t0,h0,w0
    int* rout = rinfo + BM * 4;                        // [BM] output position index or -1
    unsigned* tile_or = (unsigned*)(rout + BM);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // hardware blocks [0, full): one tile each; [full, full + rem * ksplit): K slice ks of tile full + tl.  Consecutive logical ids share an XCD,
    // so the slices of a tile mostly meet in one L2
    int bid, ks = 0, tl = 0;
    const bool split = (int)blockIdx.x >= px.full;
    if (!split) bid = xcd_remap(blockIdx.x, px.full);
    else {
        const int q = xcd_remap((int)blockIdx.x - px.full, px.rem * px.ksplit);
        tl = q / px.ksplit; ks = q % px.ksplit;
        bid = px.full + tl;
    }
    // Tile order.  Default: column tiles fastest (neighbours share the gathered rows).  mfast: ROW tiles of one (group, column tile) fastest --
    // for the spectral GEMMs, whose weight planes (1 GB for PrimaryCaps) dwarf their 320 / 448 rows per group: the row tiles that read the same
    // 3 - 4 MB of weights run side by side in one XCD's L2 instead of re-reading them from HBM (4.2 GB per launch measured for 1.1 GB of operands).
    int nt, g, lt;
    if (px.mfast) { lt = bid % p.mtiles_g; const int q = bid / p.mtiles_g; nt = q % p.ntiles; g = q / p.ntiles; }
    else { nt = bid % p.ntiles; const int mtl = bid / p.ntiles; g = mtl / p.mtiles_g; lt = mtl % p.mtiles_g; }
    const int n0 = nt * BN;
    const uint16_t* wbase = px.wp + (size_t)g * p.wgstride;
    const float* bbase = p.bias + (size_t)g * p.bgstride;
    if (tid == 0) *tile_or = 0u;

    for (int r = tid; r < BM; r += NT) {
        const int lm = lt * BM + r;
        int4 info = make_int4(-1, 0, 0, 0);
        int op = -1;
        if (lm < p.Mg) {
            int m = g * p.Mg + lm;
            int n = 0;
            if (p.flags & PC_F_NFAST) { const int ng = p.N / p.groups; n = g * ng + lm % ng; m = lm / ng; }
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

    // A pieces of this wave: piece pc = wave + NW * j covers tile rows 8 pc .. 8 pc + 7 (lane / 8), 16-byte slot lane % 8 (source swizzled)
    unsigned adma[AR], amask[AR], bdma[BR];
    const long long apos0 = ((long long)min(p.ioff0[0], 0) * p.Hi + min(p.ioff0[1], 0)) * p.Wi + min(p.ioff0[2], 0);
    const long long abias = -apos0 * p.ldi;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int row = (wave + NW * j) * 8 + (lane >> 3);
        const int4 ri = ((int4*)rinfo)[row];
        const int ks = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
        unsigned m = 0;
        if (ri.x >= 0) {
            for (int a = 0; a < p.ntap[0]; ++a) m |= ((unsigned)(ri.y + a * p.istep[0]) < (unsigned)p.Ti ? 1u : 0u) << a;
            for (int a = 0; a < p.ntap[1]; ++a) m |= ((unsigned)(ri.z + a * p.istep[1]) < (unsigned)p.Hi ? 1u : 0u) << (10 + a);
            for (int a = 0; a < p.ntap[2]; ++a) m |= ((unsigned)(ri.w + a * p.istep[2]) < (unsigned)p.Wi ? 1u : 0u) << (20 + a);
        }
        amask[j] = m;
        const long long pos = ((long long)(ri.x * p.Ti + ri.y) * p.Hi + ri.z) * p.Wi + ri.w;
        adma[j] = (unsigned)(((pos - apos0) * p.ldi + ks) * 4);
    }
    // B pieces: piece pc covers rows 16 q .. 16 q + 15 (lane / 4) of plane pl, 16-byte slot lane % 4 (source swizzled by (row >> 2) & 3)
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int pc = wave + NW * j, pl = pc / (BN / 16), q = pc % (BN / 16);
        const int row = q * 16 + (lane >> 2), co = n0 + row;
        const int sl = (lane & 3) ^ ((row >> 2) & 3);
        bdma[j] = (pc < BPIECES && co < p.Co) ? (unsigned)(((size_t)pl * px.wpstride + (size_t)co * p.wtaps * p.ldw + sl * 8) * 2) : DMA_OOB;
    }
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
    const int nb_ = b_hi - b_lo + 1, nc_ = c_hi - c_lo + 1;
    const int nchunks_tile = (a_hi - a_lo + 1) * nb_ * nc_ * (p.Ci / BK);
    // the A stream and the B stream each walk the (tap, channel chunk) sequence on their own (the prologue issues them chunk by chunk)
    struct Walk { int a, b, c, ci; };
    Walk wa = {a_lo, b_lo, c_lo, 0}, wb = wa;
    auto advance = [&](Walk& w) {
        w.ci += BK;
        if (w.ci >= p.Ci) { w.ci = 0; if (++w.c > c_hi) { w.c = c_lo; if (++w.b > b_hi) { w.b = b_lo; ++w.a; } } }
    };
    int nchunks = nchunks_tile;
    if (split) {                                                // K slice ks: chunks [c0, c1) of the tile's walk
        const int c0 = (int)((long long)nchunks_tile * ks / px.ksplit), c1 = (int)((long long)nchunks_tile * (ks + 1) / px.ksplit);
        for (int i = 0; i < c0; ++i) advance(wa);
        wb = wa;
        nchunks = c1 - c0;
    }
    auto fetchA = [&](int buf) {
        const long long da = ((long long)(wa.a * p.istep[0] * p.Hi + wa.b * p.istep[1]) * p.Wi + wa.c * p.istep[2]) * p.ldi + wa.ci;
        const unsigned sel = (1u << wa.a) | (1u << (10 + wa.b)) | (1u << (20 + wa.c));
        const dma_rsrc_t ra = dma_rsrc(p.in + (da - abias));
#pragma unroll
        for (int j = 0; j < AR; ++j) glds16b(ra, ((amask[j] & sel) == sel) ? adma[j] : DMA_OOB, (float*)(As0 + buf * ABYTES + (wave + NW * j) * 1024));
        advance(wa);
    };
    auto fetchB = [&](int buf) {
        const int wtap = ((p.wk0[0] + wb.a * p.wkstep[0]) * p.KH + p.wk0[1] + wb.b * p.wkstep[1]) * p.KW + p.wk0[2] + wb.c * p.wkstep[2];
        const dma_rsrc_t rb = dma_rsrc_h(wbase + (long long)wtap * p.ldw + wb.ci);
#pragma unroll
        for (int j = 0; j < BR; ++j)
            if (BPIECES % NW == 0 || wave + NW * j < BPIECES) glds16h(rb, bdma[j], Bs0 + buf * BBYTES + (wave + NW * j) * 1024);
        advance(wb);
    };

    // Two accumulators per tile: `hi` takes the h*h products only -- ONE rounding into the large running sum per 16 k, where an fp32 FMA chain
    // rounds 16 times -- and `lo` the five small products (2^-8 of the sum and below, so its own roundings are 2^-8 smaller).  With everything
    // in one accumulator the six accumulate steps per 16 k, which the matrix pipe does not round to nearest, left the result 1.1x further
    // from fp64 than the fp32 kernel on one-signed (ReLU) operands; split like this it is closer than the fp32 kernel on every shape tested.
    static_assert(TM == 1 && BM == 32 * WM, "one 32-row tile per wave (every configuration of pc_x6_tile)");
    // the last row tile of a group may hold fewer than BM rows (the 320-row spectral planes in 128-row tiles): a wave with no real row
    // multiplies nothing -- its share of the matrix pipe (and of the power budget the bf16 MFMAs run into) goes to the block beside it
    const bool wave_live = __builtin_amdgcn_readfirstlane(lt * BM + wm * 32 < p.Mg ? 1 : 0) != 0;
    f32x16 acc_hi[TN], acc_lo[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_hi[j][r] = 0.f; acc_lo[j][r] = 0.f; }

    // fragment addresses: step 0, buffer 0; a lane holds k = 8 (lane / 32) .. + 7 of a k16 step
    const int kh = lane >> 5;
    unsigned abase, bbas[TN];
    {
        const int r = wm * (BM / WM) + (lane & 31);
        abase = (unsigned)(r * BK * 4) + (unsigned)(((kh * 2) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / WN) + (lane & 31) + j * 32;
        bbas[j] = (unsigned)(2 * ABYTES) + (unsigned)(r * 64) + (unsigned)((kh ^ ((r >> 2) & 3)) << 4);
    }
    // Registers: A planes in two sets (the split of step t + 1 writes one while the MFMAs of step t read the other); B planes of column
    // tiles 0 .. TN - 2 in ONE set -- the planes of step t + 1 are read into a tile's registers right behind its last MFMA of step t --
    // and only the last tile's in two (its reads must be issued before the phase ends: the barrier at the head of an odd phase frees the buffer)
    f32x4 raw[2];
    u32x4 pa[2][3], pb[TN][3], pbl[2][3];
    auto rdA = [&](int half, int s, int buf) {                 // step s: slots + 4 s -> byte offset ^ 64; second half of the 8 floats: ^ 16
        raw[half] = *(const f32x4*)(smem8 + ((abase ^ (unsigned)(s * 64) ^ (unsigned)(half * 16)) + (unsigned)(buf * ABYTES)));
    };
    auto ldB = [&](int j, int pl, int s, int buf) {            // step s: plane slot kh + 2 s -> byte offset ^ 32
        return *(const u32x4*)(smem8 + ((bbas[j] ^ (unsigned)(s * 32)) + (unsigned)(buf * BBYTES + pl * BPLANE)));
    };
    auto unit = [&](int q, int set) {
        float x0 = raw[q >> 1][(q & 1) * 2], x1 = raw[q >> 1][(q & 1) * 2 + 1];
        asm volatile("" : "+v"(x0), "+v"(x1));                  // pins the unit behind the slot's sched_barrier ...
        uint32_t h, m, l;
        split2(x0, x1, h, m, l);
        asm volatile("" : "+v"(h), "+v"(m), "+v"(l));           // ... and in front of the next one
        pa[set][0][q] = h; pa[set][1][q] = m; pa[set][2][q] = l;
    };
    static_assert(NU == 4, "four split units per step");
    auto phase = [&](auto ph, int t) {
        constexpr int PH = decltype(ph)::value;
        constexpr int s = PH & 1, buf = PH >> 1, cur = PH & 1;
        constexpr int s1 = s ^ 1, b1 = s ? (buf ^ 1) : buf;     // step / buffer of t + 1
        if constexpr (s == 1) {
            // The wait is written out: with both fetches behind this barrier the only LDS-DMA in flight here comes from the PREVIOUS trip of the
            // loop, and hipcc (ROCm 7.2) then emits the barrier of __syncthreads() with lgkmcnt(0) alone -- its loop-carried bookkeeping loses the
            // pending LDS-DMA writes -- so a wave could read a tile that had not landed (caught by the tail-split test on a cold launch: 1 - 10 k
            // wrong elements in 126 of 150 launches).  Inline asm is invisible to that pass (PC_SYNC_DMA, common.h).
            PC_SYNC_DMA();                                      // chunk (t + 1) / 2 has landed; every wave has read chunk (t - 1) / 2
            if ((t + 3) / 2 < nchunks) { fetchA(buf); fetchB(buf); }       // both operands of chunk (t + 3) / 2: a whole chunk ahead of their first read
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!wave_live) return;                                 // a wave whose 32 rows lie behind the group's last row only fetches and keeps the barriers
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int j = m / 6, pr = m % 6;
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};     // hl, lh, mm, hm, mh -> lo;  hh -> hi
            if (j == TN - 1) {
                if (pr < 5) PC_MFX6(pa[cur][PA[pr]], pbl[cur][PB[pr]], acc_lo[j]);
                else PC_MFX6(pa[cur][PA[pr]], pbl[cur][PB[pr]], acc_hi[j]);
            } else {
                if (pr < 5) PC_MFX6(pa[cur][PA[pr]], pb[j][PB[pr]], acc_lo[j]);
                else PC_MFX6(pa[cur][PA[pr]], pb[j][PB[pr]], acc_hi[j]);
            }
            // LDS reads of step t + 1
            if (m < 2) rdA(m, s1, b1);
            if (m >= NM - 3) pbl[cur ^ 1][m - (NM - 3)] = ldB(TN - 1, m - (NM - 3), s1, b1);
            else if (m >= 6 && (m - 6) % 6 < 3 && (m - 6) / 6 < TN - 1) pb[(m - 6) / 6][(m - 6) % 6] = ldB((m - 6) / 6, (m - 6) % 6, s1, b1);
            // the four split units of step t + 1, spread over slots 2 ..
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (2 + (u * (NM - 2)) / 4 == m) unit(u, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;
    if (nchunks > 0) { fetchA(0); fetchB(0); }
    if (nchunks > 1) { fetchA(1); fetchB(1); }
    PC_SYNC_DMA();
    if (nchunks > 0 && wave_live) {
        rdA(0, 0, 0); rdA(1, 0, 0);
#pragma unroll
        for (int j = 0; j < TN - 1; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) pb[j][pl] = ldB(j, pl, 0, 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pbl[0][pl] = ldB(TN - 1, pl, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) unit(u, 0);
    }
    const int nph = 2 * nchunks;
    for (int t = 0; t < nph; t += 4) {
        phase(P0{}, t);
        phase(P1{}, t + 1);
        if (t + 2 < nph) {
            phase(P2{}, t + 2);
            phase(P3{}, t + 3);
        }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][j] = acc_hi[j] + acc_lo[j];
    __syncthreads();          // the operand ring is reused as the output staging tile
    if (split) {
        // partial accumulators -> workspace, in register order ([wave][j][r][lane]: 256-byte rows); the last arrival adds the slices.
        // Message passing WITHOUT fences: a device-scope release here is buffer_wbl2 -- every split block would write back its XCD's whole L2,
        // the launch's own output lines included (measured: 196 tiles in two slices each 21 -> 81 us).  Instead every slice element is stored and
        // loaded with sc0 sc1 (written through to / read from the level all XCDs share; plain buffer accesses, so 16 loads are in flight at a
        // time -- as relaxed atomics the compiler issued them one by one), the stores are waited for (vmcnt(0)) before the block's one counter
        // increment, and the loads are issued behind the barrier that publishes its result.
        const dma_rsrc_t rsw = dma_rsrc(px.ws + (size_t)tl * px.ksplit * (BM * BN));           // the tile's slices; aux 17 = sc0 sc1
        const unsigned o_own = (unsigned)((ks * (BM * BN) + wave * (TN * 16 * 64) + lane) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[0][j][r];               // (bit_cast straight from the vector element stored element 0 sixteen times)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rsw, o_own + (unsigned)((j * 16 + r) * 256), 0, 17);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *tile_or = __hip_atomic_fetch_add(px.ctr + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(px.ksplit - 1) ? 1u : 0u;
        __syncthreads();
        if (!*tile_or) return;
        const unsigned o_all = (unsigned)((wave * (TN * 16 * 64) + lane) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x16 sum;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum[r] = 0.f;
            for (int k = 0; k < px.ksplit; ++k) {               // slice order, whichever block does the adding; 16 loads in flight per slice
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    t[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsw, o_all + (unsigned)(k * (BM * BN) * 4 + (j * 16 + r) * 256), 0, 17));
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[r] += t[r];
            }
            acc[0][j] = sum;
        }
        if (tid == 0) __hip_atomic_store(px.ctr + tl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch that uses this workspace
        __syncthreads();
    }

    // ---- epilogue (as conv_gemm_glds_kernel)
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
    float* T = (float*)smem8;
    if (p.flags & PC_F_TOUT) { store_tile_cols<BM, BN, WM, WN, TM, TN, NT, EH>(acc, T, rout, rinfo, p, n0, wm, wn, lane, tid); return; }
    if (store_tile_rows<BM, BN, WM, WN, TM, TN, NT, EH>(acc, T, rout, rinfo, p, bbase, n0, wm, wn, lane, tid)) return;
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

template <int BM, int BN, int WM, int WN>
int launch_x6(const ConvX6& kx, hipStream_t s) {
    constexpr int OPBYTES = 2 * (BM * BK * 4 + 3 * BN * BK * 2);
    const size_t lds = (size_t)OPBYTES + (size_t)(BM * 5 + 4) * sizeof(int);
    PC_SET_LDS_ONCE((conv_x6_kernel<BM, BN, WM, WN>), lds, "conv_x6_kernel");
    ConvX6 p = kx;
    p.k.mtiles_g = cdiv(p.k.Mg, BM);
    p.k.ntiles = cdiv(p.k.Co, BN);
    const int tiles = p.k.groups * p.k.mtiles_g * p.k.ntiles;
    if (p.ksplit <= 1 || p.rem <= 0 || !p.ws) { p.full = tiles; p.rem = 0; p.ksplit = 1; }
    static const int mf = getenv("PICONS_X6_MFAST") ? atoi(getenv("PICONS_X6_MFAST")) : 1;
    p.mfast = mf && p.k.mtiles_g <= 8 && p.k.mtiles_g > 1 && (long long)p.k.K * BN * 6 >= (2ll << 20);     // few row tiles over >= 2 MB of weights per column tile
    const dim3 grid(p.full + p.rem * p.ksplit), block(64 * WM * WN);
    if (pc_tl_ev_start) hipExtLaunchKernelGGL((conv_x6_kernel<BM, BN, WM, WN>), grid, block, lds, s, pc_tl_ev_start, pc_tl_ev_stop, 0, p);
    else hipLaunchKernelGGL((conv_x6_kernel<BM, BN, WM, WN>), grid, block, lds, s, p);
    PC_CHECK_LAUNCH("conv_x6_kernel");
    return PC_OK;
}

// Blocks of a tile configuration that are resident on the chip at once (launch_bounds / LDS: 256 x 128 one per CU, the 4-wave tiles two, 64 x 64 three)
inline int x6_slots(const X6Tile& t) { return 256 * (t.bm == 256 ? 1 : (t.bm == 64 && t.bn == 64 ? 3 : 2)); }

// The tail split of a launch (header): which tiles are split and how far.  Only launches whose K loop is long enough to amortise the fix-up
// (>= 12 chunks per slice), and only when the last round would be less than 0.6 full.
struct X6Split { int tiles, full, rem, ksplit; long long ws_floats; };
inline X6Split x6_split(const pc_conv_desc* d, int groups, const X6Tile& t) {
    const long long Mg = (long long)(d->N / groups) * d->Tq * d->Hq * d->Wq;
    X6Split r;
    r.tiles = (int)(groups * cdiv(Mg, t.bm) * cdiv(d->Co, t.bn));
    r.full = r.tiles; r.rem = 0; r.ksplit = 1; r.ws_floats = 0;
    static const int off = getenv("PICONS_X6_TAIL_SPLIT") ? !atoi(getenv("PICONS_X6_TAIL_SPLIT")) : 0;
    if (off) return r;
    const int S = x6_slots(t);
    const int rem = r.tiles % S;
    if (rem == 0 || rem * 10 > S * 6) return r;
    const int chunks = d->ntap[0] * d->ntap[1] * d->ntap[2] * (d->Ci / BK);     // of a tile whose tap box is the whole kernel
    int ks = S / rem;
    if (ks > 8) ks = 8;
    if (ks > chunks / 12) ks = chunks / 12;
    if (ks < 2) return r;
    r.full = r.tiles - rem; r.rem = rem; r.ksplit = ks;
    r.ws_floats = (long long)rem * ks * t.bm * t.bn + ((rem + 3) / 4) * 4;        // the slices, then one counter per split tile
    return r;
}

}  // namespace

// Tile of a launch on the bf16-split kernel: the candidate with the least estimated time, time = rounds of resident blocks x the tile's
// area / its measured efficiency (256 x 128 on 8 waves: one block per CU, the fastest per multiply-accumulate; 128 x 64 and 64 x 128 on 4 waves:
// two blocks per CU; 64 x 64: three) -- the last, partial round costs as many block-times as the fullest CU holds blocks.  N-fastest row order
// (the 9 x 9 transposed / spectral forms) keeps 64-row tiles where a wider tile would straddle image rows (as conv.hip's launch_tile).
X6Tile pc_x6_tile(const pc_conv_desc* d, int groups) {
    const long long Mg = (long long)(d->N / groups) * d->Tq * d->Hq * d->Wq;
    bool rows64 = false;
    if (d->flags & PC_F_NFAST) {
        const long long per_row = (long long)d->Wq * (d->N / groups);
        rows64 = per_row % 128 != 0 && per_row % 64 == 0;
    }
    if (d->Co <= 32 && !rows64) return X6Tile{128, 32, 4, 1};
    struct Cand { X6Tile t; int bpc; double eff; };
    const Cand cand[] = {{{256, 128, 8, 1}, 1, 1.0}, {{128, 64, 4, 1}, 2, 0.9}, {{64, 128, 2, 2}, 2, 0.85}, {{64, 64, 2, 2}, 3, 0.7}};
    double best = 1e300;
    X6Tile bt = cand[3].t;
    for (const Cand& c : cand) {
        if (rows64 && c.t.bm != 64) continue;
        if (c.t.bn == 128 && d->Co <= 64) continue;
        const long long blocks = (long long)groups * cdiv(Mg, c.t.bm) * cdiv(d->Co, c.t.bn);
        const long long per_round = 256ll * c.bpc;
        const long long full = blocks / per_round, rem = blocks % per_round;
        const double t = (double)(full * c.bpc + cdiv(rem, 256)) * c.t.bm * c.t.bn / c.eff;
        if (t < best) { best = t; bt = c.t; }
    }
    return bt;
}

// A launch CAN take the bf16-split kernel when its descriptor asks for it (PC_F_X6: the caller holds weight planes) and has the shape the
// LDS-DMA tiles need: whole 32-channel chunks, at most 10 taps per dimension, 16-byte aligned plane rows.
bool pc_x6_eligible(const pc_conv_desc* d) {
    return d && (d->flags & PC_F_X6) && d->Ci % BK == 0 && d->Ci >= BK && d->ldi % 4 == 0 && d->ldw % 8 == 0 && d->ntap[0] >= 1 && d->ntap[0] <= 10 &&
           d->ntap[1] >= 1 && d->ntap[1] <= 10 && d->ntap[2] >= 1 && d->ntap[2] <= 10 && d->N % (d->groups > 0 ? d->groups : 1) == 0;
}

// Advice for the planner: eligible AND large enough to gain.  The kernel's pipeline has a longer prologue than the fp32 kernel's; launches of
// fewer than ~150 blocks of its smallest tile measured 0.75 - 0.85x (profiles/r04_x6_launches.txt) and stay on the fp32 MFMA kernel.
extern "C" int pc_conv_x6_ok(const pc_conv_desc* d) {
    if (!pc_x6_eligible(d)) return 0;
    const int groups = d->groups > 0 ? d->groups : 1;
    const X6Tile t = pc_x6_tile(d, groups);
    const long long Mg = (long long)(d->N / groups) * d->Tq * d->Hq * d->Wq;
    // ... and a K loop long enough to pay for the pipeline's prologue: launches of <= 4 chunks (K <= 128: the 1 x 1 x 1 layers over 32 - 128
    // channels) run 0.73 - 0.92x the fp32 kernel on the 4-wave tiles, 1.02 - 1.04x on the 256 x 128 tile (profiles/r04_x6_launches.txt).
    // PICONS_X6_KMIN = least number of 32-channel chunks (default 5; 1 = no rule)
    static const int kmin = getenv("PICONS_X6_KMIN") ? atoi(getenv("PICONS_X6_KMIN")) : 5;
    const int chunks = d->ntap[0] * d->ntap[1] * d->ntap[2] * (d->Ci / BK);
    if (chunks < kmin && t.bm != 256) return 0;
    return (long long)groups * cdiv(Mg, t.bm) * cdiv(d->Co, t.bn) >= 150 ? 1 : 0;
}

extern "C" int pc_split_planes(const float* src, uint16_t* planes, int64_t n, int64_t plane_stride, pc_stream s) {
    PC_CHECK_ARG(src && planes && n > 0 && n % 4 == 0 && plane_stride >= n && plane_stride % 4 == 0, "pc_split_planes: n and the plane stride must be multiples of 4 (n=%lld stride=%lld)",
                 (long long)n, (long long)plane_stride);
    PC_CHECK_ARG(((uintptr_t)src % 16 == 0) && ((uintptr_t)planes % 8 == 0), "pc_split_planes: src must be 16-byte, planes 8-byte aligned");
    const long long blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)s, src, planes, (long long)n, (long long)plane_stride);
    PC_CHECK_LAUNCH("split_planes_kernel");
    return PC_OK;
}

extern "C" int pc_split_planes_multi(const pc_split_job* jobs, int njobs, pc_stream s) {
    PC_CHECK_ARG(jobs && njobs >= 1, "pc_split_planes_multi: bad args");
    for (int j0 = 0; j0 < njobs; j0 += SP_JOBS) {
        SplitPack pk;
        pk.n = njobs - j0 < SP_JOBS ? njobs - j0 : SP_JOBS;
        long long blocks = 0;
        for (int q = 0; q < pk.n; ++q) {
            const pc_split_job& a = jobs[j0 + q];
            PC_CHECK_ARG(a.src && a.planes && a.n > 0 && a.n % 4 == 0 && a.plane_stride >= a.n && a.plane_stride % 4 == 0 && a.src % 16 == 0 && a.planes % 8 == 0,
                         "pc_split_planes_multi: bad job %d (n=%lld stride=%lld)", j0 + q, (long long)a.n, (long long)a.plane_stride);
            pk.j[q].src = (const float*)(uintptr_t)a.src; pk.j[q].dst = (uint16_t*)(uintptr_t)a.planes; pk.j[q].n = a.n; pk.j[q].pstride = a.plane_stride;
            pk.first[q] = (int)blocks;
            blocks += (a.n / 4 + 255) / 256;
            PC_CHECK_ARG(blocks < (1ll << 31), "pc_split_planes_multi: too many elements");
        }
        pk.first[pk.n] = (int)blocks;
        hipLaunchKernelGGL(split_planes_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, pk);
        PC_CHECK_LAUNCH("split_planes_multi_kernel");
    }
    return PC_OK;
}

extern "C" int64_t pc_conv_x6_ws_floats(const pc_conv_desc* d) {
    if (!pc_x6_eligible(d)) return 0;
    const int groups = d->groups > 0 ? d->groups : 1;
    return x6_split(d, groups, pc_x6_tile(d, groups)).ws_floats;
}

extern "C" int pc_conv_fwd_x6(const pc_conv_desc* d, const float* in, const uint16_t* wplanes, int64_t plane_stride, const float* bias,
                              const float* cscale, float* out, float* bnpart, pc_stream s) {
    return pc_conv_fwd_x6_ws(d, in, wplanes, plane_stride, bias, cscale, out, bnpart, nullptr, 0, s);
}

extern "C" int pc_conv_fwd_x6_ws(const pc_conv_desc* d, const float* in, const uint16_t* wplanes, int64_t plane_stride, const float* bias,
                                 const float* cscale, float* out, float* bnpart, float* ws, int64_t ws_floats, pc_stream s) {
    PC_CHECK_ARG(d && in && wplanes && out, "pc_conv_fwd_x6: null pointer");
    PC_CHECK_ARG(pc_x6_eligible(d), "pc_conv_fwd_x6: the descriptor does not take the bf16-split kernel (PC_F_X6, Ci %% 32 == 0, ldw %% 8 == 0, <= 10 taps per dimension; "
                 "Ci=%d ldi=%d ldw=%d flags=%d)", d->Ci, d->ldi, d->ldw, d->flags);
    const int groups = d->groups > 0 ? d->groups : 1;
    PC_CHECK_ARG(d->N % groups == 0, "pc_conv_fwd_x6: N %% groups");
    PC_CHECK_ARG(!(d->flags & PC_F_BIAS) || bias, "pc_conv_fwd_x6: bias flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_CSCALE) || cscale, "pc_conv_fwd_x6: cscale flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_BNPART) || bnpart, "pc_conv_fwd_x6: bnpart flag without pointer");
    PC_CHECK_ARG(!(d->flags & PC_F_NFAST) || !(d->flags & (PC_F_BNPART | PC_F_CSCALE)), "pc_conv_fwd_x6: NFAST cannot be combined with BN partials / cscale");
    PC_CHECK_ARG(!(d->flags & PC_F_TOUT) || (!(d->flags & (PC_F_BNPART | PC_F_CSCALE | PC_F_BIAS | PC_F_ACCUM | PC_F_NFAST)) && d->act == PC_ACT_NONE),
                 "pc_conv_fwd_x6: channel-major output (PC_F_TOUT) is for plain launches only");
    PC_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)wplanes % 16 == 0) && plane_stride % 8 == 0 && d->wgstride % 8 == 0,
                 "pc_conv_fwd_x6: in / wplanes must be 16-byte aligned, plane and group strides multiples of 8 elements");
    ConvX6 kx;
    ConvK& k = kx.k;
    kx.wp = wplanes; kx.wpstride = plane_stride;
    k.in = in; k.w = nullptr; k.bias = bias; k.cscale = cscale; k.out = out; k.bnpart = bnpart;
    k.N = d->N; k.Ti = d->Ti; k.Hi = d->Hi; k.Wi = d->Wi; k.Ci = d->Ci; k.ldi = d->ldi;
    k.Tq = d->Tq; k.Hq = d->Hq; k.Wq = d->Wq; k.To = d->To; k.Ho = d->Ho; k.Wo = d->Wo; k.Co = d->Co; k.ldo = d->ldo;
    for (int i = 0; i < 3; ++i) {
        k.ostr[i] = d->ostr[i]; k.ooff[i] = d->ooff[i]; k.istr[i] = d->istr[i]; k.ntap[i] = d->ntap[i];
        k.ioff0[i] = d->ioff0[i]; k.istep[i] = d->istep[i]; k.wk0[i] = d->wk0[i]; k.wkstep[i] = d->wkstep[i];
    }
    k.KH = d->KH; k.KW = d->KW; k.wtaps = d->KT * d->KH * d->KW; k.ldw = d->ldw;
    k.K = d->ntap[0] * d->ntap[1] * d->ntap[2] * d->Ci;
    const int64_t M = (int64_t)d->N * d->Tq * d->Hq * d->Wq;
    PC_CHECK_ARG(M > 0 && M < (1ll << 31) && (int64_t)d->N * d->To * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->Ti * d->Hi * d->Wi < (1ll << 31), "pc_conv_fwd_x6: position count out of range");
    k.M = (int)M; k.groups = groups; k.Mg = (int)(M / groups);
    k.act = d->act; k.flags = d->flags & ~F_SCALAR_EPI; k.act_c0 = d->act_c0; k.wgstride = d->wgstride; k.bgstride = d->bgstride;
    {   // 32-bit lane offsets from a wave-uniform base (as conv_gemm_glds_kernel)
        const long long halo = ((long long)(k.ioff0[0] < 0 ? -k.ioff0[0] : 0) * k.Hi + (k.ioff0[1] < 0 ? -k.ioff0[1] : 0)) * k.Wi + (k.ioff0[2] < 0 ? -k.ioff0[2] : 0);
        const long long in_bytes = ((long long)k.N * k.Ti * k.Hi * k.Wi + 2 * halo) * k.ldi * 4;
        const long long w_bytes = (2 * plane_stride + (long long)k.Co * k.wtaps * k.ldw) * 2;
        PC_CHECK_ARG(in_bytes < DMA_MAX_BYTES && w_bytes < DMA_MAX_BYTES,
                     "pc_conv_fwd_x6: input (%lld B) or weight planes (%lld B per group) exceed the 4 GiB the LDS-DMA gather addresses", in_bytes, w_bytes);
    }
    const X6Tile c = pc_x6_tile(d, groups);
    const X6Split sp = x6_split(d, groups, c);
    kx.full = sp.tiles; kx.rem = 0; kx.ksplit = 1; kx.ws = nullptr; kx.ctr = nullptr;
    if (ws && sp.ksplit > 1) {
        PC_CHECK_ARG(ws_floats >= sp.ws_floats && (uintptr_t)ws % 16 == 0, "pc_conv_fwd_x6_ws: the workspace holds %lld floats, this launch needs %lld (pc_conv_x6_ws_floats)",
                     (long long)ws_floats, (long long)sp.ws_floats);
        kx.full = sp.full; kx.rem = sp.rem; kx.ksplit = sp.ksplit; kx.ws = ws;
        kx.ctr = (unsigned*)(ws + (long long)sp.rem * sp.ksplit * c.bm * c.bn);        // zero before the first use; every launch leaves them zero
    }
    if (c.bm == 256) return launch_x6<256, 128, 8, 1>(kx, (hipStream_t)s);
    if (c.bm == 128 && c.bn == 64) return launch_x6<128, 64, 4, 1>(kx, (hipStream_t)s);
    if (c.bm == 128 && c.bn == 32) return launch_x6<128, 32, 4, 1>(kx, (hipStream_t)s);
    if (c.bm == 64 && c.bn == 128) return launch_x6<64, 128, 2, 2>(kx, (hipStream_t)s);
    return launch_x6<64, 64, 2, 2>(kx, (hipStream_t)s);
}
