// Capsule head for gfx950: votes + 3-iteration EM routing (forward and hand-derived reverse
// through all iterations), class-capsule masking, and the tap-sum stage of `smooth`.
// Replaces /root/reference/models/capsules_ucf101.py:290-331 (ConvCaps.forward, K=(1,1)),
// :108-211 (m_step / e_step / caps_em_routing), :438-484 (masking) and the ~150 small ATen
// launches + two 157 MB `repeat` copies per pass they cost.
//
// One wavefront per spatial position.  Lane map: h = lane&15 (pose element p*4+q),
// cg = lane>>4; the lane owns output capsules c = cg + 4*j (j < CJ).  Votes are never
// materialised: v[i][c][p,q] = sum_k P_i[p][k] * W[i][c][k][q] is recomputed from the
// position's poses and W^T held in LDS.  Reductions over h use 16-lane xor shuffles.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NB = 32;          // input capsule types (capsules_ucf101.py:355)
constexpr int MAXC = 24;        // output capsules (24 UCF101-24, 21 JHMDB-21)
constexpr int CJ = MAXC / 4;
constexpr float EPS = 1e-8f;    // capsules_ucf101.py:88
constexpr float LAMBDA = 1e-6f; // capsules_ucf101.py:90
constexpr float HALF_LN_2PI = 0.91893853320467274178f;

#define WSYNC()                                               \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

// Cross-lane sums without the LDS crossbar: `__shfl_xor` compiles to ds_bpermute_b32 (an LDS round trip of ~100 cycles plus the
// address arithmetic), and these kernels are chains of hundreds of dependent 16-lane reductions per position.  DPP row
// operations (quad_perm, row_half_mirror, row_mirror, row_ror) are modifiers of the add itself; rows of 16 lanes are combined
// with gfx950's v_permlane16_swap / v_permlane32_swap.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;                 // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;     // lane k <-> 7-k within 8 / 15-k within 16
constexpr int DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;              // rotate right within the row of 16
__device__ __forceinline__ float add_xor1(float v) { return v + dpp_mov<DPP_XOR1>(v); }
__device__ __forceinline__ float add_xor2(float v) { return v + dpp_mov<DPP_XOR2>(v); }
__device__ __forceinline__ float xor16_sum(float v) {            // v[lane] + v[lane ^ 16]
    // inline asm: with the builtin, hipcc 7.2 adds the first result to itself (v_add v, r0, r0) when both come from one value.
    // vdst's odd rows swap with src's even rows; the two s_nop cover the VALU-write -> permlane-read hazard
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor32_sum(float v) {            // v[lane] + v[lane ^ 32]
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
template <bool ROW16> __device__ __forceinline__ double swap_sum_d(double v) {   // v[lane] + v[lane ^ 16] (or ^ 32), double
    unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    unsigned lo2 = lo, hi2 = hi;
    if (ROW16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(lo), "+v"(lo2), "+v"(hi), "+v"(hi2));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(lo), "+v"(lo2), "+v"(hi), "+v"(hi2));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) + __longlong_as_double((long long)(((unsigned long long)hi2 << 32) | lo2));
}
__device__ __forceinline__ double sum_cg_d(double v) { return swap_sum_d<false>(swap_sum_d<true>(v)); }   // over cg (lane bits 4,5)
__device__ __forceinline__ float sum16(float v) {   // over h (lane bits 0..3); every lane of the row gets the sum
    v = add_xor1(v); v = add_xor2(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);                // quads are uniform by now: the mirror partner holds the other quad's sum
    v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float sum_cg(float v) {  // over cg (lane bits 4,5)
    return xor32_sum(xor16_sum(v));
}
__device__ __forceinline__ float sum_p(float v) {   // over lane bits 2,3 (the four lanes with the same low two bits of a row)
    v += dpp_mov<DPP_ROR4>(v);
    v += dpp_mov<DPP_ROR8>(v);
    return v;
}

// forward state of one position (floats).  The backward kernel replays the forward and needs every iteration's
// intermediates (FwdState); the forward kernel only needs the current one (FwdLite: 8.7 KB instead of 21.6 KB per
// wave, so ten waves instead of four share one copy of W^T in LDS).
// What the backward needs from the forward of one position, apart from the inputs themselves (5 236 floats, 21 KB).  The
// training forward kernel can leave it in global memory (pc_em_routing_fwd `state`), so the backward loads it instead of running
// the three routing iterations again (a third of its time).
struct EmSaved {
    float R[2][NB][MAXC];      // assignments entering iterations 1 and 2 (iteration 0 uses the constant 1/C)
    float invS[3][NB];
    float rs[3][MAXC];
    float mu[3][MAXC][16];
    float s2[3][MAXC][16];
    float aout[3][MAXC];
    float sd[3][MAXC][16];     // sum_i co (v - mu) as the rounded arithmetic leaves it (zero in exact arithmetic up to eps/(rs+eps))
    float D[4];                // stdv + eps per iteration
    float pad_[8];             // 5 248 floats: a whole number of 16-byte pieces
};
static_assert(sizeof(EmSaved) % 16 == 0, "EmSaved is copied in 16-byte pieces");

struct FwdState : EmSaved {
    float P[NB][16];
    float a[NB];
    float rn[NB][MAXC];        // scratch: normalised assignment of the current iteration
    static constexpr bool KEEP_SD = true;
    static __device__ __forceinline__ int ti(int t) { return t; }
    __device__ __forceinline__ float& r_prev(int t, int i, int c) { return R[t - 1][i][c]; }
    __device__ __forceinline__ float& r_next(int t, int i, int c) { return R[t][i][c]; }
    __device__ __forceinline__ float& rnorm(int i, int c) { return rn[i][c]; }
};

struct FwdLite {
    float P[NB][16];
    float a[NB];
    float R1[NB][MAXC];        // assignment of the current iteration, normalised in place
    float invS[1][NB];
    float rs[1][MAXC];
    float mu[1][MAXC][16];
    float s2[1][MAXC][16];
    float aout[1][MAXC];
    float D[4];
    static constexpr bool KEEP_SD = false;
    static __device__ __forceinline__ int ti(int) { return 0; }
    __device__ __forceinline__ float& r_prev(int, int i, int c) { return R1[i][c]; }
    __device__ __forceinline__ float& r_next(int, int i, int c) { return R1[i][c]; }
    __device__ __forceinline__ float& rnorm(int i, int c) { return R1[i][c]; }
};

struct BwdState {
    float dmu[3][MAXC][16];
    float ds2[3][MAXC][16];
    float dlnp[2][NB][MAXC];
    float dco[NB][MAXC];       // dco -> drn -> dR in place
    float da_out[MAXC];
    float drs[MAXC];
    float da_in[NB];
};

// WT layout in LDS: [i][c][q][k]  (one ds_read_b128 gives W[i][c][0..3][q])
__device__ __forceinline__ float vote(const float* WT, const f32x4 prow, int i, int c, int q, int C) {
    const f32x4 w = *(const f32x4*)(WT + ((i * C + c) * 4 + q) * 4);
    return prow[0] * w[0] + prow[1] * w[1] + prow[2] * w[2] + prow[3] * w[3];
}

// segment reductions over W consecutive lanes (W = 2 or 8, segments aligned): used to spread the per-capsule scalar
// reductions (sum over c for one i, sum over i for one c) over all threads instead of 24-32 serial lanes
template <int W> __device__ __forceinline__ float seg_sum(float v) {
    static_assert(W == 2 || W == 8, "segment width");
    v = add_xor1(v);
    if (W == 8) { v = add_xor2(v); v += dpp_mov<DPP_HALF_MIRROR>(v); }
    return v;
}
template <int W> __device__ __forceinline__ float seg_max(float v) {
    static_assert(W == 2 || W == 8, "segment width");
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    if (W == 8) { v = fmaxf(v, dpp_mov<DPP_XOR2>(v)); v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v)); }
    return v;
}

// NW cooperating waves work on ONE position: wave wv owns input capsules [wv*NB/NW, (wv+1)*NB/NW); sums over i
// are combined through `red` ([NW][MAXC*16] floats per quantity).  NW == 1: a single wave, no block barrier.
template <int NW>
__device__ __forceinline__ void SYNC() {
    if (NW == 1) WSYNC(); else __syncthreads();
}
template <class ST>
__device__ __forceinline__ float Rt(ST* st, int t, int i, int c, int C) { return t == 0 ? 1.0f / C : st->r_prev(t, i, c); }

template <int NW>
__device__ __forceinline__ void xwave_sum(float (&v)[CJ], float* red, int wv, int lane) {
    if (NW == 1) return;
    const int h = lane & 15, cg = lane >> 4;
#pragma unroll
    for (int j = 0; j < CJ; ++j) red[wv * (MAXC * 16) + (cg + 4 * j) * 16 + h] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w * (MAXC * 16) + (cg + 4 * j) * 16 + h];
        v[j] = s;
    }
    __syncthreads();
}

// Forward EM for the position held in st->P / st->a.  Leaves every iteration's state in *st.
template <int NW, class ST>
__device__ void em_forward(ST* st, const float* WT, const float* beta_u, const float* beta_a, int C, int tid, float* red, EmSaved* gs = nullptr) {
    const int lane = tid & 63, wv = tid >> 6;
    const int h = lane & 15, cg = lane >> 4, p = h >> 2, q = h & 3;
    const int i0 = wv * (NB / NW), i1 = i0 + NB / NW;
    constexpr int RP = 64 * NW / NB;           // lanes per input capsule in the row-parallel sections (2 or 8)
    constexpr int CPT = NW == 1 ? 2 : 8;       // lanes per output capsule in the column-parallel sections
    for (int t = 0; t < 3; ++t) {
        const int tt = ST::ti(t);
        // ---- M-step (capsules_ucf101.py:127-152)
        {   // S_i = sum_c r[i][c]*a_i : RP lanes per input capsule
            const int i = tid / RP, part = tid % RP;
            const float ai = st->a[i];
            float S = 0.f;
            for (int c = part; c < C; c += RP) S += Rt(st, t, i, c, C) * ai;
            S = seg_sum<RP>(S);
            if (part == 0) st->invS[tt][i] = 1.0f / (S + EPS);
        }
        SYNC<NW>();
        for (int e = tid; e < NB * C; e += 64 * NW) {
            const int i = e / C, c = e - i * C;
            st->rnorm(i, c) = Rt(st, t, i, c, C) * st->a[i] * st->invS[tt][i];
        }
        SYNC<NW>();
        if (tid < C * CPT) {   // r_sum[c] = sum_i rn[i][c] : CPT lanes per output capsule
            const int c = tid / CPT, part = tid % CPT;
            float s = 0.f;
            for (int i = part; i < NB; i += CPT) s += st->rnorm(i, c);
            s = seg_sum<CPT>(s);
            if (part == 0) st->rs[tt][c] = s;
        }
        SYNC<NW>();
        float m[CJ], sg[CJ], irs[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) { m[j] = 0.f; sg[j] = 0.f; const int c = cg + 4 * j; irs[j] = c < C ? 1.0f / (st->rs[tt][c] + EPS) : 0.f; }
        for (int i = i0; i < i1; ++i) {
            const f32x4 prow = *(const f32x4*)&st->P[i][p * 4];
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int c = cg + 4 * j;
                if (c < C) m[j] += st->rnorm(i, c) * irs[j] * vote(WT, prow, i, c, q, C);
            }
        }
        xwave_sum<NW>(m, red, wv, lane);
        float sdv[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) sdv[j] = 0.f;
        for (int i = i0; i < i1; ++i) {
            const f32x4 prow = *(const f32x4*)&st->P[i][p * 4];
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int c = cg + 4 * j;
                if (c < C) {
                    const float d = vote(WT, prow, i, c, q, C) - m[j], cw = st->rnorm(i, c) * irs[j];
                    sg[j] += cw * d * d;
                    if (ST::KEEP_SD || gs) sdv[j] += cw * d;
                }
            }
        }
        xwave_sum<NW>(sg, red, wv, lane);
        if (ST::KEEP_SD || gs) {
            // The backward needs d sigma^2 / d mu = -2 sum_i co (v - mu) with the SAME rounded differences the sigma^2 above was
            // built from: the rounding error of mu then cancels between this term and the direct 2 co (v - mu) term of each
            // vote's gradient (as it does in autograd).  Its exact-arithmetic value -2 mu eps / (rs + eps) does not cancel it,
            // and where one input capsule owns a class (v - mu -> 0, 1 / sigma^2 large) that left errors of 100 %.
            xwave_sum<NW>(sdv, red, wv, lane);
        }
        float cost[CJ]; double csum = 0.0;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = cg + 4 * j;
            cost[j] = 0.f;
            if (c < C) {
                sg[j] += EPS;
                if (wv == 0) { st->mu[tt][c][h] = m[j]; st->s2[tt][c][h] = sg[j]; if (ST::KEEP_SD) ((FwdState*)st)->sd[tt][c][h] = sdv[j]; }
                if (gs && wv == 0) { gs->mu[t][c][h] = m[j]; gs->s2[t][c][h] = sg[j]; gs->sd[t][c][h] = sdv[j]; }
                cost[j] = sum16((beta_u[c * 16 + h] + 0.5f * logf(sg[j])) * st->rs[tt][c]);
                csum += (double)cost[j];
            }
        }
        // mean and the reference's sum-then-square "stdv" (:142-144).  S = sum_c(cost - mean) is zero in exact
        // arithmetic; in the reference's fp32 it is rounding noise that moves D by a few percent (SURVEY finding 4).
        // Evaluate it in double so this path sits at the noise-free value instead of adding noise of its own.
        double cs_d = csum;
        cs_d = sum_cg_d(cs_d);
        const double mean_d = cs_d / C;
        const float mean = (float)mean_d;
        double ds_d = 0.0;
#pragma unroll
        for (int j = 0; j < CJ; ++j) if (cg + 4 * j < C) ds_d += (double)cost[j] - mean_d;
        ds_d = sum_cg_d(ds_d);
        const float D = sqrtf((float)(ds_d * ds_d / C) + EPS) + EPS;   // sum-then-square, :144
        float ao[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = cg + 4 * j;
            ao[j] = 0.f;
            if (c < C) {
                ao[j] = 1.0f / (1.0f + expf(-LAMBDA * (beta_a[c] - (mean - cost[j]) / D)));
                if (h == 0 && wv == 0) { st->aout[tt][c] = ao[j]; if (gs) gs->aout[t][c] = ao[j]; }
            }
        }
        if (tid == 0) { st->D[t] = D; if (gs) gs->D[t] = D; }
        if (gs) {
            if (tid < NB) gs->invS[t][tid] = st->invS[tt][tid];
            if (tid < C) gs->rs[t][tid] = st->rs[tt][tid];
        }
        SYNC<NW>();
        if (t == 2) break;
        // ---- E-step (capsules_ucf101.py:176-181)
        float hl[CJ], is2[CJ], la[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = cg + 4 * j;
            hl[j] = 0.f; is2[j] = 0.f; la[j] = 0.f;
            if (c < C) { hl[j] = 0.5f * logf(sg[j]) + HALF_LN_2PI; is2[j] = 1.0f / (2.0f * sg[j]); la[j] = logf(EPS + ao[j]); }
        }
        for (int i = i0; i < i1; ++i) {
            const f32x4 prow = *(const f32x4*)&st->P[i][p * 4];
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int c = cg + 4 * j;
                if (c < C) {
                    const float d = vote(WT, prow, i, c, q, C) - m[j];
                    const float lp = sum16(-d * d * is2[j] - hl[j]);
                    if (h == 0) st->r_next(t, i, c) = lp + la[j];
                }
            }
        }
        SYNC<NW>();
        {   // softmax over c, RP lanes per input capsule
            const int i = tid / RP, part = tid % RP;
            float mx = -INFINITY;
            for (int c = part; c < C; c += RP) mx = fmaxf(mx, st->r_next(t, i, c));
            mx = seg_max<RP>(mx);
            float s = 0.f;
            for (int c = part; c < C; c += RP) { const float e = expf(st->r_next(t, i, c) - mx); st->r_next(t, i, c) = e; s += e; }
            s = seg_sum<RP>(s);
            const float inv = 1.0f / s;
            for (int c = part; c < C; c += RP) { const float r = st->r_next(t, i, c) * inv; st->r_next(t, i, c) = r; if (gs) gs->R[t][i][c] = r; }
        }
        SYNC<NW>();
    }
}

__device__ __forceinline__ void load_WT(float* WT, const float* W, int C, int tid, int nthr) {
    // W [i][c][k][q] -> WT [i][c][q][k]
    for (int e = tid; e < NB * C * 16; e += nthr) {
        const int k = (e >> 2) & 3, q = e & 3;
        WT[(e & ~15) + q * 4 + k] = W[e];
    }
}

template <class ST>
__device__ __forceinline__ void load_pos(ST* st, const float* x, int64_t pos, int tid, int nthr) {
    const float* xp = x + pos * (NB * 17);
    for (int e = tid; e < NB * 16 / 4; e += nthr) ((f32x4*)&st->P[0][0])[e] = ((const f32x4*)xp)[e];
    if (tid < NB) st->a[tid] = xp[NB * 16 + tid];
}

constexpr int FWD_WAVES = 13;   // 25 positions per CU at bs = 8: 13 waves finish them in two rounds (10 or 12 need three); 13 x 8.7 KB + 48 KB of W^T = 158 KB

// Forward: one wave per position, FWD_WAVES independent waves per block sharing W^T in LDS.  The kernel is a chain
// of dependent reductions, so it lives on waves per SIMD: 4 waves/block (full FwdState) ran 1.38 ms, 8 with the
// compact state 0.74 ms, 10 0.57 ms, 12 0.59 ms (25 positions per CU: 12 leaves a nearly empty third round).
// (A cooperative 4-waves-per-position variant was measured slower: more barriers.)
__global__ __launch_bounds__(64 * FWD_WAVES) void em_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                 const float* __restrict__ beta_u, const float* __restrict__ beta_a,
                                                                 int npos, int C, float* __restrict__ out, EmSaved* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* WT = smem;
    FwdLite* sts = (FwdLite*)(smem + NB * MAXC * 16);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    load_WT(WT, W, C, threadIdx.x, blockDim.x);
    __syncthreads();
    FwdLite* st = sts + wave;
    for (int64_t pos = (int64_t)blockIdx.x * FWD_WAVES + wave; pos < npos; pos += (int64_t)gridDim.x * FWD_WAVES) {
        WSYNC();
        load_pos(st, x, pos, lane, 64);
        WSYNC();
        em_forward<1>(st, WT, beta_u, beta_a, C, lane, nullptr, state ? state + pos : nullptr);
        float* o = out + pos * (C * 17);
        for (int e = lane; e < C * 16; e += 64) o[e] = (&st->mu[0][0][0])[e];
        if (lane < C) o[C * 16 + lane] = st->aout[0][lane];
    }
}

// Backward: BWD_WAVES waves cooperate on one position at a time (forward recomputed into LDS, then reversed); a block
// holds BWD_GROUPS such teams, each with its own state, sharing one copy of W^T.  The kernel is latency-bound chains of
// reductions, so the second team is what gives every SIMD two waves; dW goes to the team's global partial with
// no-return float atomics (each element has one owner thread, so there is no contention) instead of 49 KB of LDS.
constexpr int BWD_WAVES = 4;   // 8 cooperating waves spill (256 VGPR cap at 2 waves/SIMD)
constexpr int BW = BWD_WAVES;
constexpr int BWD_GROUPS_MAX = 2;     // workspace is sized for two teams per block
constexpr int EM_SMALL = MAXC * 16 + 32;          // dbeta_u [C][16] + dbeta_a [C] accumulators

template <int BWD_GROUPS>
__global__ __launch_bounds__(64 * BWD_WAVES * BWD_GROUPS) void em_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                 const float* __restrict__ beta_u, const float* __restrict__ beta_a,
                                                                 const float* __restrict__ dout, int npos, int C, float* __restrict__ dx,
                                                                 float* __restrict__ part, const EmSaved* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 64 * BW;
    constexpr int GRP_FLOATS = EM_SMALL + (sizeof(FwdState) + sizeof(BwdState)) / 4 + BW * MAXC * 16;
    const int grp = threadIdx.x / NT, tid = threadIdx.x % NT;
    float* WT = smem;                              // [NB][C][4(q)][4(k)]  shared by the teams
    float* gbase = smem + NB * MAXC * 16 + grp * GRP_FLOATS;
    float* dbu = gbase;                            // [C][16]
    float* dba = dbu + MAXC * 16;                  // [C]
    FwdState* st = (FwdState*)(dba + 32);
    BwdState* bs = (BwdState*)(st + 1);
    float* red = (float*)(bs + 1);                 // [BW][MAXC*16]
    const int lane = tid & 63, wv = tid >> 6;
    const int h = lane & 15, cg = lane >> 4, p = h >> 2, q = h & 3;
    const int i0 = wv * (NB / BW), i1 = i0 + NB / BW;
    constexpr int RP = NT / NB, CPT = 8;
    const int team = blockIdx.x * BWD_GROUPS + grp, nteams = gridDim.x * BWD_GROUPS;
    float* pp = part + (size_t)team * (NB * MAXC * 16 + EM_SMALL);
    load_WT(WT, W, C, threadIdx.x, blockDim.x);
    for (int e = tid; e < EM_SMALL; e += NT) dbu[e] = 0.f;
    for (int e = tid; e < NB * MAXC * 16; e += NT) pp[e] = 0.f;
    __syncthreads();
    // both teams run the same number of rounds (the barriers are block-wide); a team without a position left
    // replays the last one and keeps its results to itself
    const int rounds = (npos + nteams - 1) / nteams;
    for (int rd = 0; rd < rounds; ++rd) {
        const bool live = (int64_t)team + (int64_t)rd * nteams < npos;
        const int64_t pos = live ? (int64_t)team + (int64_t)rd * nteams : npos - 1;
        __syncthreads();
        load_pos(st, x, pos, tid, NT);
        if (state) {       // the forward kernel left this position's routing state in global memory
            const f32x4* src = (const f32x4*)(state + pos);
            f32x4* dstp = (f32x4*)(EmSaved*)st;
            for (int e = tid; e < (int)(sizeof(EmSaved) / 16); e += NT) dstp[e] = src[e];
        }
        __syncthreads();
        if (!state) em_forward<BW>(st, WT, beta_u, beta_a, C, tid, red);
        // ---- seeds
        const float* dop = dout + pos * (C * 17);
        for (int e = tid; e < 3 * MAXC * 16; e += NT) { (&bs->dmu[0][0][0])[e] = 0.f; (&bs->ds2[0][0][0])[e] = 0.f; }
        __syncthreads();
        for (int e = tid; e < C * 16; e += NT) (&bs->dmu[2][0][0])[e] = dop[e];
        if (tid < C) bs->da_out[tid] = dop[C * 16 + tid];
        if (tid < NB) bs->da_in[tid] = 0.f;
        __syncthreads();
        for (int t = 2; t >= 0; --t) {
            // ---- step A: a_out / cost back to rs, beta, sigma^2  (capsules_ucf101.py:138-152); every wave computes
            // the same values, wave 0 stores them
            const float D = st->D[t];
            float du[CJ], dusum = 0.f;
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int c = cg + 4 * j;
                du[j] = 0.f;
                if (c < C) { const float ao = st->aout[t][c]; du[j] = bs->da_out[c] * ao * (1.f - ao); dusum += du[j]; }
            }
            dusum = sum_cg(dusum);            // every h-lane of a cg holds the same du, so this is the sum over c
            const float dmean = -LAMBDA / D * dusum;
            float ds_new[CJ], dmu_add[CJ], drs_new[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int c = cg + 4 * j;
                ds_new[j] = dmu_add[j] = drs_new[j] = 0.f;
                if (c < C) {
                    const float dcost = LAMBDA / D * du[j] + dmean / C;
                    const float s2v = st->s2[t][c][h], rsv = st->rs[t][c];
                    const float G = sum16(beta_u[c * 16 + h] + 0.5f * logf(s2v));
                    const float ds = bs->ds2[t][c][h] + dcost * rsv * 0.5f / s2v;
                    ds_new[j] = ds;
                    // d sigma^2 / d mu through sum_i co (v-mu)^2 :  -2 ds * sum_i co (v - mu), the sum as the forward's rounded
                    // arithmetic leaves it (see em_forward)
                    dmu_add[j] = bs->dmu[t][c][h] - 2.f * ds * st->sd[t][c][h];
                    drs_new[j] = dcost * G;
                    if (wv == 0 && live) { dbu[c * 16 + h] += dcost * rsv; if (h == 0) dba[c] += LAMBDA * du[j]; }
                }
            }
            __syncthreads();                  // all reads of ds2/dmu/da_out done before wave 0 overwrites them
            if (wv == 0) {
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int c = cg + 4 * j;
                    if (c < C) { bs->ds2[t][c][h] = ds_new[j]; bs->dmu[t][c][h] = dmu_add[j]; if (h == 0) bs->drs[c] = drs_new[j]; }
                }
            }
            // recompute rn for this iteration
            for (int e = tid; e < NB * C; e += NT) {
                const int i = e / C, c = e - i * C;
                st->rn[i][c] = Rt(st, t, i, c, C) * st->a[i] * st->invS[t][i];
            }
            __syncthreads();
            // ---- step B: dco[i][c] = sum_h dmu*v + ds2*(v-mu)^2
            {
                float muv[CJ];
#pragma unroll
                for (int j = 0; j < CJ; ++j) { const int c = cg + 4 * j; muv[j] = c < C ? st->mu[t][c][h] : 0.f; }
                for (int i = i0; i < i1; ++i) {
                    const f32x4 prow = *(const f32x4*)&st->P[i][p * 4];
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const int c = cg + 4 * j;
                        if (c < C) {
                            const float v = vote(WT, prow, i, c, q, C), d = v - muv[j];
                            const float s = sum16(dmu_add[j] * v + ds_new[j] * d * d);
                            if (h == 0) bs->dco[i][c] = s;
                        }
                    }
                }
            }
            __syncthreads();
            // ---- step C: co -> rn -> ra -> (R_t, a_in)
            if (tid < C * CPT) {
                const int c = tid / CPT, part = tid % CPT;
                const float irs = 1.0f / (st->rs[t][c] + EPS);
                float T = 0.f;
                for (int i = part; i < NB; i += CPT) T += bs->dco[i][c] * st->rn[i][c] * irs * irs;
                T = seg_sum<CPT>(T);
                if (part == 0) bs->drs[c] -= T;
            }
            __syncthreads();
            for (int e = tid; e < NB * C; e += NT) {     // dco now holds drn
                const int i = e / C, c = e - i * C;
                bs->dco[i][c] = bs->dco[i][c] / (st->rs[t][c] + EPS) + bs->drs[c];
            }
            __syncthreads();
            {                                             // ... and then dR_t (in place): RP lanes per input capsule
                const int i = tid / RP, part = tid % RP;
                const float iS = st->invS[t][i], ai = st->a[i];
                float dot = 0.f;
                for (int c = part; c < C; c += RP) dot += bs->dco[i][c] * st->rn[i][c];
                dot = seg_sum<RP>(dot);
                const float dS = -dot * iS;
                float dai = 0.f, dot2 = 0.f;
                for (int c = part; c < C; c += RP) {
                    const float dra = bs->dco[i][c] * iS + dS;
                    dai += dra * Rt(st, t, i, c, C);
                    const float dR = dra * ai;
                    bs->dco[i][c] = dR;
                    if (t > 0) dot2 += st->R[t - 1][i][c] * dR;
                }
                dai = seg_sum<RP>(dai);
                if (part == 0) bs->da_in[i] += dai;
                if (t > 0) {
                    // ---- step D(i): E-step (t-1) backward from dR_t  (capsules_ucf101.py:176-181)
                    dot2 = seg_sum<RP>(dot2);
                    for (int c = part; c < C; c += RP) bs->dlnp[t - 1][i][c] = st->R[t - 1][i][c] * (bs->dco[i][c] - dot2);
                }
            }
            __syncthreads();
            if (t == 0) break;
            if (tid < C * CPT) {
                const int c = tid / CPT, part = tid % CPT;
                float s = 0.f;
                for (int i = part; i < NB; i += CPT) s += bs->dlnp[t - 1][i][c];
                s = seg_sum<CPT>(s);
                if (part == 0) bs->da_out[c] = s / (EPS + st->aout[t - 1][c]);
            }
            {
                float am[CJ], as2[CJ], muv[CJ], is2[CJ];
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int c = cg + 4 * j;
                    am[j] = as2[j] = muv[j] = is2[j] = 0.f;
                    if (c < C) { muv[j] = st->mu[t - 1][c][h]; is2[j] = 1.0f / st->s2[t - 1][c][h]; }
                }
                for (int i = i0; i < i1; ++i) {
                    const f32x4 prow = *(const f32x4*)&st->P[i][p * 4];
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const int c = cg + 4 * j;
                        if (c < C) {
                            const float d = vote(WT, prow, i, c, q, C) - muv[j];
                            const float g = bs->dlnp[t - 1][i][c];
                            am[j] += g * d * is2[j];
                            as2[j] += g * (0.5f * d * d * is2[j] * is2[j] - 0.5f * is2[j]);
                        }
                    }
                }
                xwave_sum<BW>(am, red, wv, lane);
                xwave_sum<BW>(as2, red, wv, lane);
                if (wv == 0) {
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const int c = cg + 4 * j;
                        if (c < C) { bs->dmu[t - 1][c][h] = am[j]; bs->ds2[t - 1][c][h] = as2[j]; }
                    }
                }
            }
            __syncthreads();
        }
        // ---- final loop: total dv -> dP (to dx) and dW (LDS accumulator; wave wv owns its own i rows)
        {
            float dm[3][CJ], dsv[3][CJ], muv[3][CJ], is2[2][CJ], irs[3][CJ];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int c = cg + 4 * j;
                    dm[t][j] = dsv[t][j] = muv[t][j] = irs[t][j] = 0.f;
                    if (t < 2) is2[t][j] = 0.f;
                    if (c < C) {
                        dm[t][j] = bs->dmu[t][c][h]; dsv[t][j] = bs->ds2[t][c][h]; muv[t][j] = st->mu[t][c][h];
                        irs[t][j] = 1.0f / (st->rs[t][c] + EPS);
                        if (t < 2) is2[t][j] = 1.0f / st->s2[t][c][h];
                    }
                }
            float* dxp = dx + pos * (NB * 17);
            for (int i = i0; i < i1; ++i) {
                const f32x4 prow = *(const f32x4*)&st->P[i][p * 4];
                const float ai = st->a[i];
                f32x4 dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int c = cg + 4 * j;
                    if (c >= C) continue;   // uniform within a 16-lane group; the shuffles below stay inside the group
                    const f32x4 w = *(const f32x4*)(WT + ((i * C + c) * 4 + q) * 4);
                    const float v = prow[0] * w[0] + prow[1] * w[1] + prow[2] * w[2] + prow[3] * w[3];
                    float dv = 0.f;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const float co = Rt(st, t, i, c, C) * ai * st->invS[t][i] * irs[t][j];
                        dv += co * (dm[t][j] + 2.f * dsv[t][j] * (v - muv[t][j]));
                        if (t < 2) dv -= bs->dlnp[t][i][c] * (v - muv[t][j]) * is2[t][j];
                    }
                    dP += dv * w;
                    // dW[i][c][k][q] = sum_p P[p][k] * dv[p,q]  (reduce over p = lane bits 2,3)
                    f32x4 tk = dv * prow;
#pragma unroll
                    for (int k = 0; k < 4; ++k) tk[k] = sum_p(tk[k]);
                    const float mine = p == 0 ? tk[0] : (p == 1 ? tk[1] : (p == 2 ? tk[2] : tk[3]));
                    if (live) atomicAdd(pp + ((i * C + c) * 4 + p) * 4 + q, mine);
                }
                // dP[p][k]: reduce over q (lane bits 0,1) and cg (bits 4,5)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dP[k] = sum_cg(add_xor2(add_xor1(dP[k])));
                }
                if (q == 0 && cg == 0 && live) *(f32x4*)(dxp + i * 16 + p * 4) = dP;
            }
            if (tid < NB && live) dxp[NB * 16 + tid] = bs->da_in[tid];
        }
    }
    __syncthreads();
    for (int e = tid; e < EM_SMALL; e += NT) pp[NB * MAXC * 16 + e] = dbu[e];
}

// part [nblk][NB*MAXC*16 + MAXC*16 + 32] -> dW [NB][C][4][4], dbeta_u [C][16], dbeta_a [C]  (+=)
// A block sums 32 consecutive elements: 8 slices of the partials per element (the 32 element-threads of a slice read 128
// contiguous bytes), eight independent loads in flight per thread, slices combined through LDS in a fixed order (deterministic).
constexpr int RED_E = 32, RED_S = 8;
__global__ __launch_bounds__(256) void em_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* dW, float* dbu, float* dba) {
    __shared__ double sh[RED_S][RED_E];
    const int stride = NB * MAXC * 16 + MAXC * 16 + 32;
    const int nW = NB * C * 16, nU = C * 16;
    const int el = threadIdx.x % RED_E, sl = threadIdx.x / RED_E;
    const int e = blockIdx.x * RED_E + el;
    const bool act = e < nW + nU + C;
    int src = 0; float* dst = nullptr;
    if (act) {
        if (e < nW) { src = e; dst = dW + e; }
        else if (e < nW + nU) { src = NB * MAXC * 16 + (e - nW); dst = dbu + (e - nW); }
        else { src = NB * MAXC * 16 + MAXC * 16 + (e - nW - nU); dst = dba + (e - nW - nU); }
    }
    double s = 0.0;
    if (act) {
        int b = sl;
        for (; b + 7 * RED_S < nblk; b += 8 * RED_S) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(b + q * RED_S) * stride + src];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += (double)v[q];
        }
        for (; b < nblk; b += RED_S) s += (double)part[(size_t)b * stride + src];
    }
    sh[sl][el] = s;
    __syncthreads();
    if (sl == 0 && act) {
        double t = sh[0][el];
#pragma unroll
        for (int q = 1; q < RED_S; ++q) t += sh[q][el];
        *dst += (float)t;
    }
}

// ------------------------------------------------------------------------------ class mask
__global__ __launch_bounds__(256) void cmask_pred_kernel(const float* __restrict__ caps, int npos, int C, const float* __restrict__ cls,
                                                         const int* __restrict__ labeled, int mode, float* __restrict__ pred, float* __restrict__ mask) {
    __shared__ float sh[256];
    __shared__ float pr[MAXC];
    const int b = blockIdx.x;
    const float* cb = caps + (size_t)b * npos * C * 17 + C * 16;
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
        for (int i = threadIdx.x; i < npos; i += 256) s += cb[(size_t)i * C * 17 + c];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) { pr[c] = sh[0] / npos; pred[b * C + c] = pr[c]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int am = 0;
        for (int c = 1; c < C; ++c) if (pr[c] > pr[am]) am = c;
        const bool lab = mode != 2 && labeled[b] != 0;
        const int gt = (int)cls[b];
        for (int c = 0; c < C; ++c) {
            float m;
            if (lab) m = c == gt ? 1.f : 0.f;
            else if (mode == 0) m = 1.f;
            else m = c == am ? 1.f : 0.f;
            mask[b * C + c] = m;
        }
    }
}

__global__ __launch_bounds__(256) void cmask_apply_kernel(const float* __restrict__ caps, const float* __restrict__ mask, int npos, int C,
                                                          int64_t total, float* __restrict__ masked) {
    const int PC = C * 16;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / PC; const int e = (int)(idx - row * PC);
        const int b = (int)(row / npos);
        masked[idx] = caps[row * (C * 17) + e] * mask[b * C + (e >> 4)];
    }
}

__global__ __launch_bounds__(256) void cmask_bwd_kernel(const float* __restrict__ dmasked, const float* __restrict__ dpred, const float* __restrict__ mask,
                                                        int npos, int C, int64_t total, float* __restrict__ dcaps) {
    const int W17 = C * 17, PC = C * 16;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / W17; const int e = (int)(idx - row * W17);
        const int b = (int)(row / npos);
        float v;
        if (e < PC) v = dmasked[row * PC + e] * mask[b * C + (e >> 4)];
        else v = dpred ? dpred[b * C + (e - PC)] / npos : 0.f;
        dcaps[idx] = v;
    }
}

// ------------------------------------------------------------------------------ tap sum (smooth stage 2)
// proj [N][T][H][W][32] (27 used); ConvTranspose3d k3 p1 s1: out[o] = b + sum_k proj[o + 1 - k][k]
// Block = 8(h) x 32(w) outputs of one frame.  For each temporal tap a the 10 x 34 halo of positions is staged in
// LDS with only the 12 floats that hold taps 9a..9a+8 (coalesced float4 loads), then every thread sums its 9 taps.
constexpr int TS_H = 8, TS_W = 32, TS_LD = 13;
__global__ __launch_bounds__(256) void tapsum_fwd_kernel(const float* __restrict__ proj, int N, int T, int H, int W, const float* __restrict__ bias,
                                                         float* __restrict__ out) {
    __shared__ float slab[(TS_H + 2) * (TS_W + 2) * TS_LD];
    const int wt = (W + TS_W - 1) / TS_W, ht = (H + TS_H - 1) / TS_H;
    int bid = blockIdx.x;
    const int w0 = (bid % wt) * TS_W; bid /= wt;
    const int h0 = (bid % ht) * TS_H; bid /= ht;
    const int t = bid % T; const int n = bid / T;
    const int hl = threadIdx.x / TS_W, wl = threadIdx.x % TS_W;
    float s = bias ? bias[0] : 0.f;
    for (int a = 0; a < 3; ++a) {
        const int ti = t + 1 - a;
        if ((unsigned)ti >= (unsigned)T) continue;            // block-uniform
        const int f0 = (9 * a) / 4;                           // first float4 of the 3 that cover taps 9a..9a+8
        __syncthreads();
        for (int e = threadIdx.x; e < (TS_H + 2) * (TS_W + 2) * 3; e += 256) {
            const int pos = e / 3, f = e - pos * 3;
            const int r = pos / (TS_W + 2), c = pos - r * (TS_W + 2);
            const int hi = h0 - 1 + r, wi = w0 - 1 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                v = *(const f32x4*)(proj + ((size_t)((n * T + ti) * H + hi) * W + wi) * 32 + (f0 + f) * 4);
            float* d = slab + pos * TS_LD + f * 4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
        const int k0 = 9 * a - 4 * f0;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                s += slab[((hl + 2 - b) * (TS_W + 2) + (wl + 2 - c)) * TS_LD + k0 + 3 * b + c];
    }
    const int h = h0 + hl, w = w0 + wl;
    if (h < H && w < W) out[(size_t)((n * T + t) * H + h) * W + w] = s;
}

// dproj[i][k] = dout[i - 1 + k]
__global__ __launch_bounds__(256) void tapsum_bwd_kernel(const float* __restrict__ dout, int N, int T, int H, int W, float* __restrict__ dproj) {
    const int64_t total = (int64_t)N * T * H * W * 8;   // float4 groups of the 32 channels
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx >> 3; const int g4 = (int)(idx & 7);
        const int64_t pos = r;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T); const int n = (int)(r / T);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tap = g4 * 4 + e;
            float v = 0.f;
            if (tap < 27) {
                const int a = tap / 9, bb = (tap / 3) % 3, c = tap % 3;
                const int to = t - 1 + a, ho = h - 1 + bb, wo = w - 1 + c;
                if ((unsigned)to < (unsigned)T && (unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W)
                    v = dout[(size_t)((n * T + to) * H + ho) * W + wo];
            }
            o[e] = v;
        }
        *(f32x4*)(dproj + pos * 32 + g4 * 4) = o;
    }
}

inline int em_bwd_blocks(int npos, int groups = BWD_GROUPS_MAX) { const int b = (npos + groups - 1) / groups; return b < 256 ? b : 256; }
// PICONS_EM_GROUPS: teams per block of the EM backward.  2 = 154 KB of LDS per block: nothing else fits on the CU while it runs;
// 1 = 102 KB: one 128x64 GEMM block of another lane (49 KB) can share the CU, so the skip-conv / weight-gradient lanes keep moving.
inline int em_groups() { static const int g = getenv("PICONS_EM_GROUPS") ? atoi(getenv("PICONS_EM_GROUPS")) : 2; return g == 1 ? 1 : 2; }
constexpr size_t EM_PART = NB * MAXC * 16 + MAXC * 16 + 32;

}  // namespace

extern "C" int64_t pc_em_ws_floats(int npos, int B, int C) {
    (void)B; (void)C;
    return (int64_t)256 * BWD_GROUPS_MAX * (int64_t)EM_PART + 64;
}

extern "C" int64_t pc_em_state_floats(int npos) { return (int64_t)npos * (int64_t)(sizeof(EmSaved) / 4); }

extern "C" int pc_em_routing_fwd(const float* x, const float* W, const float* beta_u, const float* beta_a, int npos, int B, int C,
                                 float* out, float* state, pc_stream s) {
    PC_CHECK_ARG(x && W && beta_u && beta_a && out, "pc_em_routing_fwd: null");
    PC_CHECK_ARG(!state || (uintptr_t)state % 16 == 0, "pc_em_routing_fwd: state alignment");
    PC_CHECK_ARG(B == NB && C >= 1 && C <= MAXC, "pc_em_routing_fwd: B must be 32 and C <= 24 (B=%d C=%d)", B, C);
    PC_CHECK_ARG((uintptr_t)x % 16 == 0, "pc_em_routing_fwd: x alignment");
    const size_t lds = (size_t)NB * MAXC * 16 * 4 + sizeof(FwdLite) * FWD_WAVES;
    PC_SET_LDS_ONCE(em_fwd_kernel, lds, "em_fwd_kernel");
    int grid = cdiv(npos, FWD_WAVES);
    if (grid > 256) grid = 256;            // one block (10 waves, 136 KB of LDS) per CU, positions strided over the grid
    hipLaunchKernelGGL(em_fwd_kernel, dim3(grid), dim3(64 * FWD_WAVES), lds, (hipStream_t)s, x, W, beta_u, beta_a, npos, C, out, (EmSaved*)state);
    PC_CHECK_LAUNCH("em_fwd");
    return PC_OK;
}

extern "C" int pc_em_routing_bwd(const float* x, const float* W, const float* beta_u, const float* beta_a, const float* dout, int npos,
                                 int B, int C, float* dx, float* dW, float* dbeta_u, float* dbeta_a, float* ws, const float* state, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(x && W && beta_u && beta_a && dout && dx && dW && dbeta_u && dbeta_a && ws, "pc_em_routing_bwd: null");
    PC_CHECK_ARG(!state || (uintptr_t)state % 16 == 0, "pc_em_routing_bwd: state alignment");
    PC_CHECK_ARG(B == NB && C >= 1 && C <= MAXC, "pc_em_routing_bwd: B must be 32 and C <= 24");
    const int BWD_GROUPS = em_groups();
    const size_t lds = (size_t)NB * MAXC * 16 * 4 + BWD_GROUPS * ((size_t)EM_SMALL * 4 + sizeof(FwdState) + sizeof(BwdState) + BWD_WAVES * MAXC * 16 * 4);
    PC_SET_LDS_ONCE(em_bwd_kernel<1>, lds, "em_bwd_kernel<1>");
    PC_SET_LDS_ONCE(em_bwd_kernel<2>, lds, "em_bwd_kernel<2>");
    const int nblk = em_bwd_blocks(npos, BWD_GROUPS);
    if (BWD_GROUPS == 1) hipLaunchKernelGGL(em_bwd_kernel<1>, dim3(nblk), dim3(64 * BWD_WAVES), lds, s, x, W, beta_u, beta_a, dout, npos, C, dx, ws, (const EmSaved*)state);
    else hipLaunchKernelGGL(em_bwd_kernel<2>, dim3(nblk), dim3(64 * BWD_WAVES * 2), lds, s, x, W, beta_u, beta_a, dout, npos, C, dx, ws, (const EmSaved*)state);
    PC_CHECK_LAUNCH("em_bwd");
    hipLaunchKernelGGL(em_reduce_kernel, dim3(cdiv(NB * C * 16 + C * 17, RED_E)), dim3(256), 0, s, ws, nblk * BWD_GROUPS, C, dW, dbeta_u, dbeta_a);
    PC_CHECK_LAUNCH("em_reduce");
    return PC_OK;
}

extern "C" int pc_class_mask_fwd(const float* caps, int Bn, int npos_per_b, int C, const float* cls, const int32_t* labeled, int mode,
                                 float* actor_pred, float* mask, float* masked, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(caps && actor_pred && mask && masked && C <= MAXC, "pc_class_mask_fwd: bad args");
    PC_CHECK_ARG(mode == 2 || (cls && labeled), "pc_class_mask_fwd: cls/labeled required in train mode");
    hipLaunchKernelGGL(cmask_pred_kernel, dim3(Bn), dim3(256), 0, s, caps, npos_per_b, C, cls, labeled, mode, actor_pred, mask);
    const int64_t total = (int64_t)Bn * npos_per_b * C * 16;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(cmask_apply_kernel, dim3(grid), dim3(256), 0, s, caps, mask, npos_per_b, C, total, masked);
    PC_CHECK_LAUNCH("class_mask_fwd");
    return PC_OK;
}

extern "C" int pc_class_mask_bwd(const float* dmasked, const float* dactor_pred, const float* mask, int Bn, int npos_per_b, int C,
                                 float* dcaps, pc_stream s) {
    PC_CHECK_ARG(dmasked && mask && dcaps, "pc_class_mask_bwd: null");
    const int64_t total = (int64_t)Bn * npos_per_b * C * 17;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(cmask_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, dmasked, dactor_pred, mask, npos_per_b, C, total, dcaps);
    PC_CHECK_LAUNCH("class_mask_bwd");
    return PC_OK;
}

extern "C" int pc_tapsum_fwd(const float* proj, int N, int T, int H, int W, const float* bias, float* out, pc_stream s) {
    PC_CHECK_ARG(proj && out, "pc_tapsum_fwd: null");
    const int64_t grid = (int64_t)N * T * ((H + TS_H - 1) / TS_H) * ((W + TS_W - 1) / TS_W);
    PC_CHECK_ARG(grid < (1ll << 31), "pc_tapsum_fwd: grid too large");
    hipLaunchKernelGGL(tapsum_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, proj, N, T, H, W, bias, out);
    PC_CHECK_LAUNCH("tapsum_fwd");
    return PC_OK;
}

extern "C" int pc_tapsum_bwd(const float* dout, int N, int T, int H, int W, float* dproj, pc_stream s) {
    PC_CHECK_ARG(dout && dproj, "pc_tapsum_bwd: null");
    const int64_t total = (int64_t)N * T * H * W * 8;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(tapsum_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, dout, N, T, H, W, dproj);
    PC_CHECK_LAUNCH("tapsum_bwd");
    return PC_OK;
}
