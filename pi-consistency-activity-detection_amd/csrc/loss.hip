// Fused consistency/supervised loss for gfx950 (HBM-bound, two streaming passes).
// Replaces, on device and in one call, what the reference does with 2 host round-trips and
// ~10 ATen reductions per step: /root/reference/main_ucf101.py:89-148 +
// utils/helpers.py:8-67 (bv mask), :70-95 (gv mask), utils/losses.py:44-57,74-76.
// Thread = one (clip, h, w) column of the 8 frames; block partials -> tiny finalize kernels.
#include "common.h"
#include <stdlib.h>
#include <utility>

namespace {

constexpr int T8 = 8;
enum { Q_MINC, Q_MAXC, Q_MINA, Q_MAXA, Q_MING, Q_MAXG, Q_SSQ, Q_AC, Q_AA, Q_BCE, Q_INTER, Q_SSIG, Q_SSEG, Q_N };
// clipstat [B][16]: 0 minC 1 invC 2 minA 3 invA 4 minG 5 invG(f32 semantics) ; glob [16]

struct LossK {
    const float* O; const float* F; const float* seg; const int* labeled;
    int B, H, W, HW; int bv, gv, n_frames, predict_maps;
    float lower, upper;
    double* part;        // [nbx][B][Q_N]
    double* clip;        // [B][16]
    double* glob;        // [16]
    double* gvpart;      // [nbx2]
    float c_l2, c_bv, c_gv, wt_loc, wt_cons;
    float* dO; float* dF; float* mask_bv; float* mask_gv; float* scalars;
    int nbx, nbx2;
};

// np.var of the N = 2*HALF+1 cyclic neighbours of element T of a 14-frame sequence, in numpy's float32 order (sequential sum,
// mean, sequential sum of squared deviations).  Every index is a compile-time constant: the sequence stays in registers.
template <int HALF, int T>
__device__ __forceinline__ float np_var14(const float (&x)[14]) {
    constexpr int N = 2 * HALF + 1;
    float s = x[(T - HALF + 14) % 14];
#pragma unroll
    for (int k = 1; k < N; ++k) s += x[(T + k - HALF + 14) % 14];
    const float mean = s / (float)N;
    float d = x[(T - HALF + 14) % 14] - mean, acc = d * d;
#pragma unroll
    for (int k = 1; k < N; ++k) { d = x[(T + k - HALF + 14) % 14] - mean; acc += d * d; }
    return acc / (float)N;
}
template <int HALF, int... T>
__device__ __forceinline__ void var14_all(const float (&x)[14], float (&V)[14], std::integer_sequence<int, T...>) {
    ((V[T] = np_var14<HALF, T>(x)), ...);
}

// raw (un-normalised) folded cyclic variance of helpers.py:25-57 for both call sites of
// main_ucf101.py:114-115.  o[8], fp[8] -> Mc[8] (clockwise), Ma[8] (anticlockwise, NOT yet time-flipped)
template <int HALF>
__device__ __forceinline__ void var_raw_t(const float* o, const float* fp, double* Mc, double* Ma) {
    float cyc[14], cya[14];
#pragma unroll
    for (int t = 0; t < 8; ++t) { cyc[t] = o[t]; cya[t] = o[7 - t]; }
#pragma unroll
    for (int m = 0; m < 6; ++m) { cyc[8 + m] = fp[6 - m]; cya[8 + m] = fp[1 + m]; }
    float Vc[14], Va[14];
    var14_all<HALF>(cyc, Vc, std::make_integer_sequence<int, 14>{});
    var14_all<HALF>(cya, Va, std::make_integer_sequence<int, 14>{});
    Mc[0] = 2.0 * (double)Vc[0]; Mc[7] = 2.0 * (double)Vc[7];
    Ma[0] = 2.0 * (double)Va[0]; Ma[7] = 2.0 * (double)Va[7];
#pragma unroll
    for (int k = 1; k < 7; ++k) { Mc[k] = (double)Vc[k] + (double)Vc[14 - k]; Ma[k] = (double)Va[k] + (double)Va[14 - k]; }
}
__device__ __forceinline__ void var_raw(const float* o, const float* fp, int half, double* Mc, double* Ma) {
    if (half == 2) var_raw_t<2>(o, fp, Mc, Ma);       // --n_frames 5
    else var_raw_t<1>(o, fp, Mc, Ma);                 // --n_frames 3
}

__device__ __forceinline__ void grad2_raw(const float* o, float lower, float upper, float* g) {
    float s[8], g1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float v = 1.0f / (1.0f + expf(-o[t]));
        if (lower >= 0.f && v < lower) v = 0.f;
        if (upper >= 0.f && v > upper) v = 1.f;
        s[t] = v;
    }
    g1[0] = s[1] - s[0]; g1[7] = s[7] - s[6];
#pragma unroll
    for (int t = 1; t < 7; ++t) g1[t] = (s[t + 1] - s[t - 1]) / 2.0f;
    g[0] = g1[1] - g1[0]; g[7] = g1[7] - g1[6];
#pragma unroll
    for (int t = 1; t < 7; ++t) g[t] = (g1[t + 1] - g1[t - 1]) / 2.0f;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double block_min(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
}

__device__ __forceinline__ void load_col(const LossK& p, int b, int hw, float* o, float* fp, float* sg, bool need_seg) {
    const int h = hw / p.W, w = hw - h * p.W;
    const size_t base = (size_t)b * T8 * p.HW;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        o[t] = p.O[base + (size_t)t * p.HW + hw];
        fp[t] = p.F[base + (size_t)t * p.HW + h * p.W + (p.W - 1 - w)];    // torch.flip(flip_op,[4]), main_ucf101.py:100
        if (need_seg) sg[t] = p.seg[base + (size_t)t * p.HW + hw];
    }
}

// NC adjacent columns (w0 .. w0+NC-1 of one image row) per thread: NC = 4 moves 16 bytes per lane and load / store
// instruction (W % 4 == 0 and 16-byte aligned tensors), NC = 1 is the general form.  The flipped operand's columns
// W-1-w0 .. W-NC-w0 are one aligned vector too, read in reverse.
template <int NC>
__device__ __forceinline__ void load_cols(const LossK& p, int b, int hw0, float (&o)[NC][8], float (&fp)[NC][8], float (&sg)[NC][8], bool need_fp, bool need_seg) {
    if (NC == 1) {
        float fdummy[8];
        load_col(p, b, hw0, o[0], need_fp ? fp[0] : fdummy, sg[0], need_seg);
        return;
    }
    typedef float vecT __attribute__((ext_vector_type(NC)));
    const int h = hw0 / p.W, w0 = hw0 - h * p.W;
    const size_t base = (size_t)b * T8 * p.HW;
    const int fl = h * p.W + (p.W - NC - w0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const vecT v = *(const vecT*)(p.O + base + (size_t)t * p.HW + hw0);
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c][t] = v[c];
        if (need_fp) {
            const vecT f = *(const vecT*)(p.F + base + (size_t)t * p.HW + fl);
#pragma unroll
            for (int c = 0; c < NC; ++c) fp[c][t] = f[NC - 1 - c];
        }
        if (need_seg) {
            const vecT g = *(const vecT*)(p.seg + base + (size_t)t * p.HW + hw0);
#pragma unroll
            for (int c = 0; c < NC; ++c) sg[c][t] = g[c];
        }
    }
}

template <int NC> struct LossGeom { static constexpr int BT = NC == 1 ? 256 : 128; };   // threads per block

// ---- pass 1: per-clip partial statistics.  One wave-level reduction per quantity, ONE block barrier for all thirteen.
template <int NC>
__global__ __launch_bounds__(256) void loss_pass1(const LossK p) {
    constexpr int BT = LossGeom<NC>::BT, NW = BT / 64;
    __shared__ double sh[NW][Q_N];
    const int b = blockIdx.y, hw0 = (blockIdx.x * BT + threadIdx.x) * NC;
    const bool act = hw0 < p.HW;
    const bool lab = p.labeled[b] != 0;
    double q[Q_N];
    q[Q_MINC] = q[Q_MINA] = q[Q_MING] = 1e300; q[Q_MAXC] = q[Q_MAXA] = q[Q_MAXG] = -1e300;
    for (int k = Q_SSQ; k < Q_N; ++k) q[k] = 0.0;
    if (act) {
        float oa[NC][8], fa[NC][8], sa[NC][8];
        load_cols<NC>(p, b, hw0, oa, fa, sa, true, lab);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* o = oa[c]; const float* fp = fa[c]; const float* sg = sa[c];
            float sq[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float d = fp[t] - o[t]; sq[t] = d * d; q[Q_SSQ] += (double)sq[t]; }
            if (p.bv) {
                float oo[8], ff[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { oo[t] = p.predict_maps ? 1.0f / (1.0f + expf(-o[t])) : o[t]; ff[t] = p.predict_maps ? 1.0f / (1.0f + expf(-fp[t])) : fp[t]; }
                double Mc[8], Ma[8];
                var_raw(oo, ff, p.n_frames / 2, Mc, Ma);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    q[Q_MINC] = fmin(q[Q_MINC], Mc[t]); q[Q_MAXC] = fmax(q[Q_MAXC], Mc[t]);
                    q[Q_MINA] = fmin(q[Q_MINA], Ma[t]); q[Q_MAXA] = fmax(q[Q_MAXA], Ma[t]);
                    q[Q_AC] += (double)sq[t] * Mc[t];
                    q[Q_AA] += (double)sq[t] * Ma[7 - t];      // torch.flip(batch_variance_anticlck,[2]), main_ucf101.py:121
                }
            }
            if (p.gv) {
                float g[8];
                grad2_raw(o, p.lower, p.upper, g);
#pragma unroll
                for (int t = 0; t < 8; ++t) { q[Q_MING] = fmin(q[Q_MING], (double)g[t]); q[Q_MAXG] = fmax(q[Q_MAXG], (double)g[t]); }
            }
            if (lab) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float x = o[t], y = sg[t];
                    const float s = 1.0f / (1.0f + expf(-x));
                    // BCEWithLogits: max(x,0) - x*y + log(1+exp(-|x|))
                    q[Q_BCE] += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
                    q[Q_INTER] += (double)(s * y); q[Q_SSIG] += (double)s; q[Q_SSEG] += (double)y;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < Q_N; ++k) {
        double v = q[k];
        const bool is_min = k == Q_MINC || k == Q_MINA || k == Q_MING, is_max = k == Q_MAXC || k == Q_MAXA || k == Q_MAXG;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double u = __shfl_xor(v, o, 64);
            v = is_min ? fmin(v, u) : (is_max ? fmax(v, u) : v + u);
        }
        if (lane == 0) sh[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < Q_N) {
        const int k = threadIdx.x;
        const bool is_min = k == Q_MINC || k == Q_MINA || k == Q_MING, is_max = k == Q_MAXC || k == Q_MAXA || k == Q_MAXG;
        double v = sh[0][k];
        for (int w = 1; w < NW; ++w) v = is_min ? fmin(v, sh[w][k]) : (is_max ? fmax(v, sh[w][k]) : v + sh[w][k]);
        p.part[((size_t)blockIdx.x * p.B + b) * Q_N + k] = v;
    }
}

// ---- mid: reduce the partials of ONE clip per block to its statistics.  The partials come from other CUs (memory latency ~2 us
// per dependent round trip), so every thread owns one quantity and a slice of the x blocks and issues its loads together.
// clip[b][0..5]: min / 1/range of the two variance masks, min / range of the gradient mask; clip[b][8..15]: the clip's share of
// the eight global sums (read back by loss_globals in pass 2 and the final kernel).
constexpr int MID_SL = 16;           // x slices per quantity: 13 * 16 = 208 of the 256 threads
__global__ __launch_bounds__(256) void loss_mid(const LossK p) {
    __shared__ double qs[MID_SL][Q_N];
    const int b = blockIdx.x;
    const int k = threadIdx.x % Q_N, sl = threadIdx.x / Q_N;
    const bool is_min = k == Q_MINC || k == Q_MINA || k == Q_MING, is_max = k == Q_MAXC || k == Q_MAXA || k == Q_MAXG;
    if (sl < MID_SL) {
        double v = is_min ? 1e300 : (is_max ? -1e300 : 0.0);
        const double* src = p.part + (size_t)b * Q_N + k;
        const size_t stride = (size_t)p.B * Q_N;
        int x = sl;
        for (; x + 3 * MID_SL < p.nbx; x += 4 * MID_SL) {
            const double u0 = src[(size_t)x * stride], u1 = src[(size_t)(x + MID_SL) * stride], u2 = src[(size_t)(x + 2 * MID_SL) * stride],
                         u3 = src[(size_t)(x + 3 * MID_SL) * stride];
            v = is_min ? fmin(fmin(v, u0), fmin(u1, fmin(u2, u3))) : (is_max ? fmax(fmax(v, u0), fmax(u1, fmax(u2, u3))) : (((v + u0) + u1) + u2) + u3);
        }
        for (; x < p.nbx; x += MID_SL) { const double u = src[(size_t)x * stride]; v = is_min ? fmin(v, u) : (is_max ? fmax(v, u) : v + u); }
        qs[sl][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < Q_N) {
        double v = qs[0][k];
        for (int i = 1; i < MID_SL; ++i) v = is_min ? fmin(v, qs[i][k]) : (is_max ? fmax(v, qs[i][k]) : v + qs[i][k]);
        qs[0][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double* q = qs[0];
        double* cs = p.clip + b * 16;
        // helpers.py:59-61: M -= min; M /= (max(M) - min(M) + 1e-7) on the shifted array (min = 0)
        const double rc = (q[Q_MAXC] - q[Q_MINC]) + 1e-7, ra = (q[Q_MAXA] - q[Q_MINA]) + 1e-7;
        cs[0] = q[Q_MINC]; cs[1] = 1.0 / rc; cs[2] = q[Q_MINA]; cs[3] = 1.0 / ra;
        // helpers.py:88-89 in float32
        const float mg = (float)q[Q_MING], xg = (float)q[Q_MAXG];
        const float rg = ((xg - mg) - 0.0f) + 1e-7f;
        cs[4] = (double)mg; cs[5] = (double)rg;
        const bool lab = p.labeled[b] != 0;
        cs[8] = q[Q_SSQ];
        cs[9] = p.bv ? (q[Q_AC] - q[Q_MINC] * q[Q_SSQ]) / rc : 0.0;
        cs[10] = p.bv ? (q[Q_AA] - q[Q_MINA] * q[Q_SSQ]) / ra : 0.0;
        cs[11] = lab ? q[Q_BCE] : 0.0; cs[12] = lab ? q[Q_INTER] : 0.0; cs[13] = lab ? q[Q_SSIG] : 0.0; cs[14] = lab ? q[Q_SSEG] : 0.0;
        cs[15] = lab ? 1.0 : 0.0;
    }
}

// the eight global sums (ssq, var-weighted clockwise / anticlockwise, bce, inter, ssig, sseg, labeled clips) from the clips' shares,
// summed in clip order by every caller
__device__ __forceinline__ double loss_global(const LossK& p, int which) {
    double s = 0.0;
    for (int i = 0; i < p.B; ++i) s += p.clip[i * 16 + 8 + which];
    return s;
}

// ---- pass 2: gradients (+ optional mask outputs, gv cross-sample sum).  Thread = NC columns of ONE clip; under --gv the
// cross-sample weight sum_j (g_j - min_j) / range_j of utils/losses.py:74-76's (B, B, ...) broadcast is rebuilt per thread from
// the B clips' columns (they sit in L2 / Infinity Cache: 12.8 MB), in the same j order as a serial loop.
template <int NC>
__global__ __launch_bounds__(256) void loss_pass2(const LossK p) {
    constexpr int BT = LossGeom<NC>::BT, NW = BT / 64;
    __shared__ double sh[NW];
    const int b = blockIdx.y, hw0 = (blockIdx.x * BT + threadIdx.x) * NC;
    const bool act = hw0 < p.HW;
    const double NT = (double)T8 * p.HW;
    const double BN = (double)p.B * NT;
    const double nl = loss_global(p, 7);
    const double inter = loss_global(p, 4), ssig = loss_global(p, 5), sseg = loss_global(p, 6);
    const double den = ssig + sseg + 1.0;
    double gvacc = 0.0;
    if (act) {
        const bool lab = p.labeled[b] != 0;
        float gsum[NC][8];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < 8; ++t) gsum[c][t] = 0.f;
        if (p.gv) {
            for (int j = 0; j < p.B; ++j) {
                float oj[NC][8], fj[NC][8], sj[NC][8];
                load_cols<NC>(p, j, hw0, oj, fj, sj, false, false);
                const float mg = (float)p.clip[j * 16 + 4], rg = (float)p.clip[j * 16 + 5];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float g[8];
                    grad2_raw(oj[c], p.lower, p.upper, g);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float gn = (g[t] - mg) / rg;
                        gsum[c][t] += gn;
                        if (p.mask_gv && j == b) p.mask_gv[((size_t)j * T8 + t) * p.HW + hw0 + c] = gn;
                    }
                }
            }
        }
        float oa[NC][8], fa[NC][8], sa[NC][8];
        load_cols<NC>(p, b, hw0, oa, fa, sa, true, lab);
        float goa[NC][8], gca[NC][8];
        const double* cs = p.clip + b * 16;
        const size_t base = (size_t)b * T8 * p.HW;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* o = oa[c]; const float* fp = fa[c]; const float* sg = sa[c];
            double Mc[8], Ma[8];
            if (p.bv) {
                float oo[8], ff[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { oo[t] = p.predict_maps ? 1.0f / (1.0f + expf(-o[t])) : o[t]; ff[t] = p.predict_maps ? 1.0f / (1.0f + expf(-fp[t])) : fp[t]; }
                var_raw(oo, ff, p.n_frames / 2, Mc, Ma);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float d = fp[t] - o[t];
                double wgt = (double)p.c_l2 / BN;
                if (p.bv) {
                    const float mc = (float)((Mc[t] - cs[0]) * cs[1]);
                    const float ma = (float)((Ma[7 - t] - cs[2]) * cs[3]);
                    wgt += (double)p.c_bv * ((double)mc + (double)ma) / BN;
                    if (p.mask_bv) p.mask_bv[base + (size_t)t * p.HW + hw0 + c] = mc;
                }
                if (p.gv) {
                    wgt += (double)p.c_gv * (double)gsum[c][t] / (BN * p.B);
                    gvacc += (double)gsum[c][t] * (double)(d * d);
                }
                const float gc = (float)(2.0 * (double)d * wgt * (double)p.wt_cons);
                float go = -gc;
                if (lab) {
                    const float x = o[t], y = sg[t];
                    const float s = 1.0f / (1.0f + expf(-x));
                    const double dbce = ((double)s - (double)y) / (nl * NT);
                    const double ddice = -(2.0 * (double)y * den - (2.0 * inter + 1.0)) / (den * den) * (double)(s * (1.f - s));
                    go += (float)((double)p.wt_loc * (dbce + ddice));
                }
                goa[c][t] = go; gca[c][t] = gc;
            }
        }
        const int h = hw0 / p.W, w0 = hw0 - h * p.W;
        if (NC == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                p.dO[base + (size_t)t * p.HW + hw0] = goa[0][t];
                p.dF[base + (size_t)t * p.HW + h * p.W + (p.W - 1 - w0)] = gca[0][t];
            }
        } else {
            typedef float vecT __attribute__((ext_vector_type(NC)));
            const int fl = h * p.W + (p.W - NC - w0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                vecT vo, vf;
#pragma unroll
                for (int c = 0; c < NC; ++c) { vo[c] = goa[c][t]; vf[NC - 1 - c] = gca[c][t]; }
                *(vecT*)(p.dO + base + (size_t)t * p.HW + hw0) = vo;
                *(vecT*)(p.dF + base + (size_t)t * p.HW + fl) = vf;
            }
        }
    }
    gvacc = wave_sum_d(gvacc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = gvacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int w = 0; w < NW; ++w) r += sh[w];
        p.gvpart[(size_t)b * p.nbx + blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void loss_final(const LossK p) {
    __shared__ double sh[4];
    double gv = 0.0;
    for (int i = threadIdx.x; i < p.nbx2; i += 256) gv += p.gvpart[i];
    gv = block_sum(gv, sh);
    if (threadIdx.x != 0) return;
    const double NT = (double)T8 * p.HW, BN = (double)p.B * NT;
    const double nl = loss_global(p, 7);
    const double l2 = loss_global(p, 0) / BN;
    const double lv = (loss_global(p, 1) + loss_global(p, 2)) / BN;
    const double lg = gv / (BN * p.B);
    const double bce = nl > 0 ? loss_global(p, 3) / (nl * NT) : 0.0;
    const double dice = 1.0 - (2.0 * loss_global(p, 4) + 1.0) / (loss_global(p, 5) + loss_global(p, 6) + 1.0);
    const double cons = (double)p.c_l2 * l2 + (double)p.c_bv * lv + (double)p.c_gv * lg;
    p.scalars[0] = (float)(bce + dice); p.scalars[1] = (float)cons; p.scalars[2] = (float)bce; p.scalars[3] = (float)dice;
    p.scalars[4] = (float)l2; p.scalars[5] = (float)lv; p.scalars[6] = (float)lg; p.scalars[7] = (float)nl;
}

// standalone mask kernels (utils.helpers drop-ins): reuse pass1/mid with a degenerate setup
__global__ __launch_bounds__(256) void var_mask_write(const LossK p, float* mask, int which) {
    const int b = blockIdx.y, hw = blockIdx.x * 256 + threadIdx.x;
    if (hw >= p.HW) return;
    float o[8], fp[8];
    const size_t base = (size_t)b * T8 * p.HW;
#pragma unroll
    for (int t = 0; t < 8; ++t) { o[t] = p.O[base + (size_t)t * p.HW + hw]; fp[t] = p.F[base + (size_t)t * p.HW + hw]; }
    if (p.predict_maps) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { o[t] = 1.0f / (1.0f + expf(-o[t])); fp[t] = 1.0f / (1.0f + expf(-fp[t])); }
    }
    (void)which;
    double Mc[8], Ma[8];
    // measure_pixelwise_var_v2(pred, flip_pred): cyc = cat(pred[0:8], flip_pred[1:7]); var_raw's clockwise
    // arm builds cat(o, fp[6..1]) so feed it the time-reversed flip_pred.
    float fr[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) fr[t] = fp[7 - t];
    var_raw(o, fr, p.n_frames / 2, Mc, Ma);
    const double* cs = p.clip + b * 16;
#pragma unroll
    for (int t = 0; t < 8; ++t) mask[base + (size_t)t * p.HW + hw] = (float)((Mc[t] - cs[0]) * cs[1]);
}

__global__ __launch_bounds__(256) void var_mask_stats(const LossK p) {
    __shared__ double sh[4];
    const int b = blockIdx.y, hw = blockIdx.x * 256 + threadIdx.x;
    double mn = 1e300, mx = -1e300;
    if (hw < p.HW) {
        float o[8], fp[8], fr[8];
        const size_t base = (size_t)b * T8 * p.HW;
#pragma unroll
        for (int t = 0; t < 8; ++t) { o[t] = p.O[base + (size_t)t * p.HW + hw]; fp[t] = p.F[base + (size_t)t * p.HW + hw]; }
        if (p.predict_maps) {
#pragma unroll
            for (int t = 0; t < 8; ++t) { o[t] = 1.0f / (1.0f + expf(-o[t])); fp[t] = 1.0f / (1.0f + expf(-fp[t])); }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) fr[t] = fp[7 - t];
        double Mc[8], Ma[8];
        var_raw(o, fr, p.n_frames / 2, Mc, Ma);
#pragma unroll
        for (int t = 0; t < 8; ++t) { mn = fmin(mn, Mc[t]); mx = fmax(mx, Mc[t]); }
    }
    double* out = p.part + ((size_t)blockIdx.x * p.B + b) * Q_N;
    const double a = block_min(mn, sh), c = -block_min(-mx, sh);
    if (threadIdx.x == 0) {
        for (int k = 0; k < Q_N; ++k) out[k] = 0.0;
        out[Q_MINC] = a; out[Q_MAXC] = c; out[Q_MINA] = 0; out[Q_MAXA] = 0; out[Q_MING] = 0; out[Q_MAXG] = 0;
    }
}

__global__ __launch_bounds__(256) void grad_mask_stats(const LossK p) {
    __shared__ double sh[4];
    const int b = blockIdx.y, hw = blockIdx.x * 256 + threadIdx.x;
    double mn = 1e300, mx = -1e300;
    if (hw < p.HW) {
        float o[8], g[8];
        const size_t base = (size_t)b * T8 * p.HW;
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = p.O[base + (size_t)t * p.HW + hw];
        grad2_raw(o, p.lower, p.upper, g);
#pragma unroll
        for (int t = 0; t < 8; ++t) { mn = fmin(mn, (double)g[t]); mx = fmax(mx, (double)g[t]); }
    }
    double* out = p.part + ((size_t)blockIdx.x * p.B + b) * Q_N;
    const double a = block_min(mn, sh), c = -block_min(-mx, sh);
    if (threadIdx.x == 0) {
        for (int k = 0; k < Q_N; ++k) out[k] = 0.0;
        out[Q_MING] = a; out[Q_MAXG] = c;
    }
}

__global__ __launch_bounds__(256) void grad_mask_write(const LossK p, float* mask) {
    const int b = blockIdx.y, hw = blockIdx.x * 256 + threadIdx.x;
    if (hw >= p.HW) return;
    float o[8], g[8];
    const size_t base = (size_t)b * T8 * p.HW;
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = p.O[base + (size_t)t * p.HW + hw];
    grad2_raw(o, p.lower, p.upper, g);
    const float mg = (float)p.clip[b * 16 + 4], rg = (float)p.clip[b * 16 + 5];
#pragma unroll
    for (int t = 0; t < 8; ++t) mask[base + (size_t)t * p.HW + hw] = (g[t] - mg) / rg;
}

// ---- SpreadLoss (utils/losses.py:14-37) on labeled rows; single block
__global__ __launch_bounds__(64) void spread_kernel(const float* __restrict__ x, const float* __restrict__ cls, const int* __restrict__ labeled,
                                                    int Bn, int C, float m, float wt, float* out, float* dx) {
    __shared__ float sl[64], sa[64];
    __shared__ int nlab;
    if (threadIdx.x == 0) { int n = 0; for (int i = 0; i < Bn; ++i) n += labeled[i] != 0; nlab = n; }
    __syncthreads();
    const int b = nlab;
    float l = 0.f, al = 0.f;
    for (int e = threadIdx.x; e < Bn * C; e += 64) {
        const int i = e / C, c = e - i * C;
        if (!labeled[i]) continue;
        const float at = x[i * C + (int)cls[i]];
        const float v = m - (at - x[e]), va = 0.9f - (at - x[e]);
        if (v > 0.f) l += v * v;
        if (va > 0.f) al += va * va;
    }
    sl[threadIdx.x] = l; sa[threadIdx.x] = al;
    __syncthreads();
    if (threadIdx.x == 0) {
        float L = 0.f, A = 0.f;
        for (int i = 0; i < 64; ++i) { L += sl[i]; A += sa[i]; }
        out[0] = b > 0 ? (L / b - m * m) / b : 0.f;        // divides by b twice (:34-35)
        out[1] = b > 0 ? A / b - 0.81f : 0.f;
    }
    if (dx && b > 0) {
        // d loss / d x[i][c] = 2*max(0, m - at + x_c)/b^2 for c != t ; for c == t: -sum_{c'!=t} 2*max(..)/b^2 (own term has zero slope)
        for (int i = threadIdx.x; i < Bn; i += 64) {
            if (!labeled[i]) continue;
            const int tcl = (int)cls[i];
            const float at = x[i * C + tcl];
            float acc = 0.f;
            for (int c = 0; c < C; ++c) {
                if (c == tcl) continue;
                const float v = m - (at - x[i * C + c]);
                const float g = v > 0.f ? 2.f * v / ((float)b * (float)b) : 0.f;
                dx[i * C + c] += wt * g;
                acc += g;
            }
            dx[i * C + tcl] -= wt * acc;
        }
    }
}

inline void fill_lossk(LossK& k, const pc_loss_desc* d) {
    k.B = d->B; k.H = d->H; k.W = d->W; k.HW = d->H * d->W;
    k.bv = d->bv; k.gv = d->gv; k.n_frames = d->n_frames; k.predict_maps = d->predict_maps;
    k.lower = d->lower_thresh; k.upper = d->upper_thresh;
    k.nbx = cdiv(k.HW, 256); k.nbx2 = k.nbx * k.B;
}
inline void carve_ws(LossK& k, float* ws) {
    double* w = (double*)ws;
    k.part = w; w += (size_t)k.nbx * k.B * Q_N;
    k.clip = w; w += (size_t)k.B * 16;
    k.glob = w; w += 16;
    k.gvpart = w;   // nbx * B + 32 doubles (the standalone mask entries borrow it as a zeroed labeled[] array)
}

}  // namespace

extern "C" int64_t pc_loss_ws_floats(const pc_loss_desc* d) {
    const int64_t nbx = cdiv((int64_t)d->H * d->W, 256);
    return 2 * (nbx * d->B * Q_N + d->B * 16 + 16 + nbx * d->B + 32) + 64;
}

extern "C" int pc_consistency_loss(const pc_loss_desc* d, const float* output, const float* flip_op, const float* seg,
                                   const int32_t* labeled, float* scalars, float* d_output, float* d_flip_op, float* mask_bv,
                                   float* mask_gv, float* ws, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(d && output && flip_op && seg && labeled && scalars && d_output && d_flip_op && ws, "pc_consistency_loss: null");
    PC_CHECK_ARG(d->T == T8, "pc_consistency_loss: T must be 8 (utils/helpers.py:14), got %d", d->T);
    PC_CHECK_ARG(d->B >= 1 && d->B <= 64, "pc_consistency_loss: B out of range");
    PC_CHECK_ARG(!d->bv || d->n_frames == 3 || d->n_frames == 5, "pc_consistency_loss: n_frames must be 3 or 5 (helpers.py:35-47)");
    PC_CHECK_ARG((uintptr_t)ws % 8 == 0, "pc_consistency_loss: ws alignment");
    LossK k;
    fill_lossk(k, d);
    k.O = output; k.F = flip_op; k.seg = seg; k.labeled = labeled;
    carve_ws(k, ws);
    k.dO = d_output; k.dF = d_flip_op; k.mask_bv = mask_bv; k.mask_gv = mask_gv; k.scalars = scalars;
    k.wt_loc = d->wt_loc; k.wt_cons = d->wt_cons;
    // main_ucf101.py:136-143 / main_jhmdb.py:121,132
    const float r = d->wt_ramp;
    if (d->jhmdb) {
        if (d->gv) { k.c_l2 = 0; k.c_bv = 0; k.c_gv = 1; }
        else if (d->bv) { k.c_l2 = 1 - r; k.c_bv = r; k.c_gv = 0; }
        else { k.c_l2 = 1; k.c_bv = 0; k.c_gv = 0; }
    } else if (d->bv && d->gv) { k.c_l2 = d->bv_wt * (1 - r); k.c_bv = d->bv_wt * r; k.c_gv = d->gv_wt; }
    else if (d->gv) { k.c_l2 = 0; k.c_bv = 0; k.c_gv = 1; }
    else if (d->bv) { k.c_l2 = 1 - r; k.c_bv = r; k.c_gv = 0; }
    else { k.c_l2 = 1; k.c_bv = 0; k.c_gv = 0; }
    // four columns per thread (16-byte loads / stores) when a row is a whole number of them and every tensor is 16-byte aligned
    static const int nc_env = getenv("PICONS_LOSS_NC") ? atoi(getenv("PICONS_LOSS_NC")) : 1;      // columns per thread: 1 measured fastest (89 / 101 / 116 us for 1 / 2 / 4: the passes are bound by the variance arithmetic, not by bytes, and wider threads cost occupancy)
    const int nc_env2 = nc_env;
    const bool vec4 = nc_env >= 2 && d->W % 4 == 0 && (((uintptr_t)output | (uintptr_t)flip_op | (uintptr_t)seg | (uintptr_t)d_output | (uintptr_t)d_flip_op) & 15) == 0;
    if (vec4 && nc_env2 == 2) {
        k.nbx = cdiv(k.HW / 2, LossGeom<2>::BT); k.nbx2 = k.nbx * k.B;         // = the cdiv(HW, 256) the workspace is sized for
        hipLaunchKernelGGL(loss_pass1<2>, dim3(k.nbx, k.B), dim3(LossGeom<2>::BT), 0, s, k);
        hipLaunchKernelGGL(loss_mid, dim3(k.B), dim3(256), 0, s, k);
        hipLaunchKernelGGL(loss_pass2<2>, dim3(k.nbx, k.B), dim3(LossGeom<2>::BT), 0, s, k);
    } else if (vec4) {
        k.nbx = cdiv(k.HW / 4, LossGeom<4>::BT); k.nbx2 = k.nbx * k.B;         // <= the cdiv(HW, 256) the workspace is sized for
        hipLaunchKernelGGL(loss_pass1<4>, dim3(k.nbx, k.B), dim3(LossGeom<4>::BT), 0, s, k);
        hipLaunchKernelGGL(loss_mid, dim3(k.B), dim3(256), 0, s, k);
        hipLaunchKernelGGL(loss_pass2<4>, dim3(k.nbx, k.B), dim3(LossGeom<4>::BT), 0, s, k);
    } else {
        hipLaunchKernelGGL(loss_pass1<1>, dim3(k.nbx, k.B), dim3(256), 0, s, k);
        hipLaunchKernelGGL(loss_mid, dim3(k.B), dim3(256), 0, s, k);
        hipLaunchKernelGGL(loss_pass2<1>, dim3(k.nbx, k.B), dim3(256), 0, s, k);
    }
    hipLaunchKernelGGL(loss_final, dim3(1), dim3(256), 0, s, k);
    PC_CHECK_LAUNCH("consistency_loss");
    return PC_OK;
}

extern "C" int pc_var_mask(const float* pred, const float* flip_pred, int B, int T, int H, int W, int n_frames, int use_sig,
                           float* mask, float* ws, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(pred && flip_pred && mask && ws && T == T8 && B >= 1 && B <= 64, "pc_var_mask: bad args");
    PC_CHECK_ARG(n_frames == 3 || n_frames == 5, "pc_var_mask: n_frames must be 3 or 5");
    pc_loss_desc d = {}; d.B = B; d.T = T; d.H = H; d.W = W; d.bv = 1; d.n_frames = n_frames; d.predict_maps = use_sig;
    d.lower_thresh = -1; d.upper_thresh = -1;
    LossK k; fill_lossk(k, &d); k.O = pred; k.F = flip_pred; k.seg = nullptr; k.labeled = nullptr; carve_ws(k, ws);
    hipLaunchKernelGGL(var_mask_stats, dim3(k.nbx, B), dim3(256), 0, s, k);
    // reuse loss_mid for the reduction: it reads labeled[] -> give it a zeroed region of ws (clip area is written after reads)
    k.labeled = (const int*)(k.gvpart);   // gvpart is unused here; zero it first
    (void)hipMemsetAsync((void*)k.gvpart, 0, sizeof(double) * (k.nbx + 32), s);
    hipLaunchKernelGGL(loss_mid, dim3(k.B), dim3(256), 0, s, k);
    hipLaunchKernelGGL(var_mask_write, dim3(k.nbx, B), dim3(256), 0, s, k, mask, 0);
    PC_CHECK_LAUNCH("var_mask");
    return PC_OK;
}

extern "C" int pc_grad_mask(const float* pred, int B, int T, int H, int W, float lower, float upper, float* mask, float* ws, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(pred && mask && ws && T == T8 && B >= 1 && B <= 64, "pc_grad_mask: bad args");
    pc_loss_desc d = {}; d.B = B; d.T = T; d.H = H; d.W = W; d.gv = 1; d.lower_thresh = lower; d.upper_thresh = upper;
    LossK k; fill_lossk(k, &d); k.O = pred; k.F = pred; k.seg = nullptr; carve_ws(k, ws);
    hipLaunchKernelGGL(grad_mask_stats, dim3(k.nbx, B), dim3(256), 0, s, k);
    k.labeled = (const int*)(k.gvpart);
    (void)hipMemsetAsync((void*)k.gvpart, 0, sizeof(double) * (k.nbx + 32), s);
    hipLaunchKernelGGL(loss_mid, dim3(k.B), dim3(256), 0, s, k);
    hipLaunchKernelGGL(grad_mask_write, dim3(k.nbx, B), dim3(256), 0, s, k, mask);
    PC_CHECK_LAUNCH("grad_mask");
    return PC_OK;
}

extern "C" int pc_spread_loss(const float* x, const float* cls, const int32_t* labeled, int Bn, int C, float m, float wt, float* out,
                              float* dx, pc_stream s) {
    PC_CHECK_ARG(x && cls && labeled && out, "pc_spread_loss: null");
    hipLaunchKernelGGL(spread_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, x, cls, labeled, Bn, C, m, wt, out, dx);
    PC_CHECK_LAUNCH("spread_loss");
    return PC_OK;
}
