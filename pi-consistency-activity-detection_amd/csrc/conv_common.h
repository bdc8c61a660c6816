// Pieces shared by the gather-GEMM conv kernels (conv.hip: fp32 MFMA; conv_x6.hip: fp32 emulated on the bf16 matrix cores):
// the kernel-side descriptor, the LDS-DMA helpers and the output-tile epilogues.
#pragma once
#include "common.h"

namespace {

constexpr int F_SCALAR_EPI = 1 << 29;     // internal flag (PICONS_CONV_SCALAR_EPI=1): 4-byte stores straight from the accumulators

struct ConvK {
    const float* in; const float* w; const float* bias; const float* cscale; float* out; float* bnpart;
    int N, Ti, Hi, Wi, Ci, ldi;
    int Tq, Hq, Wq, To, Ho, Wo, Co, ldo;
    int ostr[3], ooff[3], istr[3], ntap[3], ioff0[3], istep[3], wk0[3], wkstep[3];
    int KH, KW, wtaps, ldw;
    int K, M, Mg, groups, mtiles_g, ntiles;
    int act, flags, act_c0, wgstride, bgstride;
};

constexpr int BK = 32;       // K chunk (floats)
constexpr int CONV_DEFAULT_VARIANT = 0;
constexpr int LDK = 36;      // padded LDS row (floats): conflict-free ds_read_b128 (9i mod 16 distinct)

// Output tile through LDS (the operand buffers are free once the K loop has ended on a barrier) so that every lane stores
// 16 bytes of one row and a wave covers whole 128..512-byte row segments.  The accumulator layout (one column, 16 rows
// per lane) gives 4-byte stores, 64 per thread, whose drain is NOT hidden behind the other resident block's MFMAs
// (0.10 ms of the 0.31 ms K = 128 tail GEMM).  Needs 4-column granularity of the output; returns false otherwise.
template <int BM, int BN, int WM, int WN, int TM, int TN, int NT = 256, int EH = 1>
__device__ __forceinline__ bool store_tile_rows(const f32x16 (&acc)[TM][TN], float* T, const int* rout, const int* rinfo, const ConvK& p,
                                                const float* bbase, int n0, int wm, int wn, int lane, int tid) {
    // EH > 1: the staging area holds BM / EH rows; the wave rows go through it in EH passes (the 256-row tile of conv_x6.hip)
    static_assert(WM % EH == 0, "epilogue passes split the wave rows");
    constexpr int RH = BM / EH;
    const bool has_bias = p.flags & PC_F_BIAS, has_cs = p.flags & PC_F_CSCALE, accum = p.flags & PC_F_ACCUM;
    if (!(((p.Co | p.ldo) & 3) == 0 && ((uintptr_t)p.out & 15) == 0 && !(p.flags & F_SCALAR_EPI) && (!has_bias || ((uintptr_t)bbase & 15) == 0) &&
          (!has_cs || ((uintptr_t)p.cscale & 15) == 0)))
        return false;
#pragma unroll
    for (int h = 0; h < EH; ++h) {
        if (EH == 1 || wm / (WM / EH) == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) - h * RH;
#pragma unroll
                    for (int j = 0; j < TN; ++j) T[row * BN + wn * (BN / WN) + j * 32 + (lane & 31)] = acc[i][j][r];
                }
        }
        __syncthreads();
        for (int e = tid; e < RH * BN / 4; e += NT) {
            const int row = e / (BN / 4) + h * RH, c4 = e % (BN / 4);
            const int op = rout[row], col = n0 + c4 * 4;
            if (op < 0 || col >= p.Co) continue;
            f32x4 v = *(const f32x4*)(T + (row - h * RH) * BN + c4 * 4);
            if (has_bias) v += *(const f32x4*)(bbase + col);
            if (p.act != PC_ACT_NONE) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col + q >= p.act_c0) {
                        if (p.act == PC_ACT_RELU) v[q] = fmaxf(v[q], 0.f);
                        else if (p.act == PC_ACT_SIGMOID) v[q] = 1.0f / (1.0f + expf(-v[q]));
                    }
            }
            if (has_cs) v *= *(const f32x4*)(p.cscale + (size_t)rinfo[row * 4] * p.Co + col);
            float* o = p.out + (size_t)op * p.ldo + col;
            if (accum) v += *(const f32x4*)o;
            *(f32x4*)o = v;
        }
        if (h + 1 < EH) __syncthreads();
    }
    return true;
}

// Channel-major output (PC_F_TOUT): the tile goes through LDS like store_tile_rows, but is read back column by column so that
// a wave writes 64 consecutive positions of ONE output channel (256 contiguous bytes where the tile's rows are consecutive
// positions).  The 16-byte column groups of a row are XOR-ed with the row so both the accumulator-layout writes (32 consecutive
// columns of one row per half-wave) and the column reads (32 consecutive rows of one column) hit distinct banks.
template <int BM, int BN, int WM, int WN, int TM, int TN, int NT = 256, int EH = 1>
__device__ __forceinline__ void store_tile_cols(const f32x16 (&acc)[TM][TN], float* T, const int* rout, const int* rinfo, const ConvK& p,
                                                int n0, int wm, int wn, int lane, int tid) {
    static_assert(WM % EH == 0, "epilogue passes split the wave rows");
    constexpr int RH = BM / EH;
    const size_t P3 = (size_t)p.To * p.Ho * p.Wo;
#pragma unroll
    for (int h = 0; h < EH; ++h) {
        if (EH == 1 || wm / (WM / EH) == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) - h * RH;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = wn * (BN / WN) + j * 32 + (lane & 31);
                        T[row * BN + (col ^ (row & 31))] = acc[i][j][r];
                    }
                }
        }
        __syncthreads();
        for (int e = tid; e < RH * BN; e += NT) {
            const int col = e / RH, lrow = e % RH, row = lrow + h * RH;
            const int op = rout[row];
            if (op < 0 || n0 + col >= p.Co) continue;
            const size_t n = (size_t)rinfo[row * 4];
            p.out[(n * p.ldo + n0 + col) * P3 + ((size_t)op - n * P3)] = T[lrow * BN + (col ^ (lrow & 31))];
        }
        if (h + 1 < EH) __syncthreads();
    }
}

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The same 1 KiB LDS-DMA piece through a raw buffer resource: wave-uniform base (SGPRs, rebuilt per K chunk with scalar adds) + one
// 32-bit byte offset per lane.  Measured beside the MFMA stream (tools/mfma_peak_probe.hip, PROBE_DMA=1): a piece with 64-bit per-lane
// addresses costs ~60 cycles of matrix-pipe time (the address VGPR pairs and the 64-bit adds that make them), this form ~24.  A lane
// whose offset is DMA_OOB lies outside the resource: the hardware writes ZEROS into its LDS slot (tools/dma_oob_probe.hip), so padding
// taps need no zero line and no address select.  Tensors addressed this way must be smaller than 4 GiB (checked by the launchers).
constexpr unsigned DMA_OOB = 0xffffffffu;
constexpr long long DMA_MAX_BYTES = 0xff000000ll;
typedef __amdgpu_buffer_rsrc_t dma_rsrc_t;
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffff00u, 0x00020000);
}
__device__ __forceinline__ void glds16b(dma_rsrc_t rs, unsigned voff, float* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}


// two fp32 -> three packed bf16 pairs (x0 in the low half), round-to-nearest-even at every level: h = bf16(x), r = x - h (exact),
// m = bf16(r), l = r - m (exact: at most 8 significant bits, so bf16(l) == l).  h + m + l == x bit for bit for |x| >= 2^-110 (below, the
// lower terms are subnormal differences, which the vector ALU flushes) and |x| < 2^127 * 1.99 (above, h rounds to infinity).  Rounding to
// nearest -- not truncation, which costs the same five instructions per pair and level -- makes the residuals signed: the three dropped
// products m*l', l*m', l*l' (weight 2^-25 and below) then carry no common sign and do not bias the sum (measured on ReLU outputs:
// truncation 1.2x the fp32 FMA chain's error against fp64, rounding 0.9x).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));      // opaque to the compiler, which otherwise converts the low element a second time
    return r;
}
__device__ __forceinline__ void x6_split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

}  // namespace

// fp32 emulated on the bf16 matrix cores (conv_x6.hip): which launches take that kernel and with which tile -- ONE classification for
// pc_conv_fwd_x6, pc_conv_bnpart_rows and the host-side work accounting (pc_conv_work)
struct X6Tile { int bm, bn, wm, wn; };
bool pc_x6_eligible(const pc_conv_desc* d);
X6Tile pc_x6_tile(const pc_conv_desc* d, int groups);
