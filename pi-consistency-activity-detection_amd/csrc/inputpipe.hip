// Input pipeline on the device: what /root/reference/datasets/ucf_dataloader.py does per sample on
// the host in numpy after decoding -- pick 8 frames, crop 224x224, /255, the horizontally flipped copy, and the
// foreground mask from the per-frame boxes (:146-173, :204-221) -- from the uint8 frames already in HBM, writing the fp32
// NCDHW staging tensors the step engine consumes.  HBM-bound: 3 bytes read, 28 bytes written per pixel.
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct ClipK {
    const uint8_t* video; int F, H, W;
    int span[8]; int h0, w0, S;          // frame ids, crop origin, crop size (224)
    const int32_t* rects; int R;         // [8][R][4] = x0, x1, y0, y1 in frame coordinates (empty: x1 <= x0)
    const uint8_t* maskf; int valid[8]; float* mask_cls;   // JHMDB form: per-pixel truth frames [F][H][W] (> 0 = foreground), frames that carry truth
    float* data; float* aug; float* mask;   // [3][8][S][S], [3][8][S][S], [8][S][S]
    int nhwc4;                              // 1: data / aug as [8][S][S][4] (r, g, b, 0): the layout the network's first conv reads
};

__global__ __launch_bounds__(256) void clip_from_u8_kernel(const ClipK p) {
    const int S = p.S;
    const int64_t total = (int64_t)8 * S * S;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int w = (int)(idx % S), h = (int)((idx / S) % S), t = (int)(idx / ((int64_t)S * S));
        const int f = p.span[t], y = h + p.h0, x = w + p.w0;
        const uint8_t* px = p.video + (((size_t)f * p.H + y) * p.W + x) * 3;
        if (p.nhwc4) {
            float4 v;
            v.x = (float)((double)px[0] / 255.0); v.y = (float)((double)px[1] / 255.0); v.z = (float)((double)px[2] / 255.0); v.w = 0.f;
            const size_t row = ((size_t)t * S + h) * S;
            ((float4*)p.data)[row + w] = v;
            ((float4*)p.aug)[row + (S - 1 - w)] = v;
        } else
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)((double)px[c] / 255.0);          // img / 255. in float64, then the caller's float32 cast
            const size_t o = (((size_t)c * 8 + t) * S + h) * S;
            p.data[o + w] = v;
            p.aug[o + (S - 1 - w)] = v;                               // video_rgb[:, :, ::-1, :]
        }
        float m = 0.f;
        if (p.maskf) {
            if (p.valid[t] && p.maskf[((size_t)f * p.H + y) * p.W + x] > 0) m = 1.f;
            p.mask_cls[idx] = p.valid[t] ? 1.f : 0.f;
        } else {
            const int32_t* r = p.rects + (size_t)t * p.R * 4;
            for (int q = 0; q < p.R; ++q)
                if (x >= r[q * 4] && x < r[q * 4 + 1] && y >= r[q * 4 + 2] && y < r[q * 4 + 3]) m = 1.f;
        }
        p.mask[idx] = m;
    }
}

}  // namespace

static int clip_launch(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S, const int32_t* rects, int R,
                       const uint8_t* maskf, const int32_t* valid8, float* data, float* aug, float* mask, float* mask_cls, pc_stream s, int nhwc4 = 0);

extern "C" int pc_clip_from_u8(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S,
                               const int32_t* rects, int R, float* data, float* aug, float* mask, pc_stream s) {
    return clip_launch(video, F, H, W, span8, h0, w0, S, rects, R, nullptr, nullptr, data, aug, mask, nullptr, s);
}

extern "C" int pc_clip_from_u8_masks(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S,
                                     const uint8_t* maskframes, const int32_t* valid8, float* data, float* aug, float* mask, float* mask_cls,
                                     pc_stream s) {
    PC_CHECK_ARG(maskframes && valid8 && mask_cls, "pc_clip_from_u8_masks: null pointer");
    return clip_launch(video, F, H, W, span8, h0, w0, S, nullptr, 0, maskframes, valid8, data, aug, mask, mask_cls, s);
}

extern "C" int pc_clip_from_u8_ndhwc4(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S,
                                      const int32_t* rects, int R, float* data, float* aug, float* mask, pc_stream s) {
    PC_CHECK_ARG(((uintptr_t)data % 16 == 0) && ((uintptr_t)aug % 16 == 0), "pc_clip_from_u8_ndhwc4: data / aug must be 16-byte aligned");
    return clip_launch(video, F, H, W, span8, h0, w0, S, rects, R, nullptr, nullptr, data, aug, mask, nullptr, s, 1);
}

static int clip_launch(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S, const int32_t* rects, int R,
                       const uint8_t* maskf, const int32_t* valid8, float* data, float* aug, float* mask, float* mask_cls, pc_stream s, int nhwc4) {
    PC_CHECK_ARG(video && span8 && data && aug && mask && (rects || R == 0), "pc_clip_from_u8: null pointer");
    PC_CHECK_ARG(F >= 1 && S >= 1 && h0 >= 0 && w0 >= 0 && h0 + S <= H && w0 + S <= W && R >= 0, "pc_clip_from_u8: crop %d+%d x %d+%d outside %d x %d", h0, S, w0, S, H, W);
    ClipK k;
    k.video = video; k.F = F; k.H = H; k.W = W; k.h0 = h0; k.w0 = w0; k.S = S; k.rects = rects; k.R = R; k.data = data; k.aug = aug; k.mask = mask;
    k.maskf = maskf; k.mask_cls = mask_cls; k.nhwc4 = nhwc4;
    for (int t = 0; t < 8; ++t) k.valid[t] = valid8 ? valid8[t] : 0;
    for (int t = 0; t < 8; ++t) {
        PC_CHECK_ARG(span8[t] >= 0 && span8[t] < F, "pc_clip_from_u8: frame %d outside [0, %d)", span8[t], F);
        k.span[t] = span8[t];
    }
    const int64_t total = (int64_t)8 * S * S;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(clip_from_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, k);
    PC_CHECK_LAUNCH("clip_from_u8");
    return PC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// cv2.resize on uint8 images, the interpolations the reference's loaders call: INTER_AREA for the JHMDB frames
// (datasets/jhmdb_dataloader.py:252, 320x240 -> 256x256), INTER_NEAREST for its puppet masks (:267,:281), INTER_LINEAR for
// a 224 crop resized to another frame size (:192,:208; ucf_dataloader.py:165,171).  OpenCV is not in this image, so what is
// restated here is its published algorithm (modules/imgproc/src/resize.cpp, 4.x): the coordinate / coefficient tables are
// built on the host in the same double / float arithmetic (pc_resize_tables), the kernels do the integer (fixed-point,
// INTER_RESIZE_COEF_BITS = 11) or sequential float32 work per output pixel.
//   kind 0 NEAREST   sx = min(floor(dx * (1 / (Wo / W))), W - 1)
//   kind 1 LINEAR    two taps per axis, 11-bit coefficients, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2;
//                    also what INTER_AREA does when either axis enlarges (coefficients from the cell overlap: the JHMDB case,
//                    x shrinks 1.25x, y grows 1.0667x)
//   kind 2 AREA      both axes shrink, fractional cell coverage, float32 accumulation in table order
//   kind 3 AREA_FAST both scales integer: window sum; 2x2: (sum + 2) >> 2, else round-half-even of sum * (1.f / area);
//                    also INTER_LINEAR at exactly 2x2 decimation
//   kind 4 COPY      same size
namespace {

enum { RK_NEAREST = 0, RK_LINEAR = 1, RK_AREA = 2, RK_AREA_FAST = 3, RK_COPY = 4 };
constexpr int RHDR = 8;          // header words: kind, off_x, off_xa, off_y, off_ya, nx / xmax / iscale_x, ny / iscale_y, float-fx offset

inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }
inline int32_t fbits(float f) { int32_t b; memcpy(&b, &f, 4); return b; }
inline short sat_short(float v) { long r = lrintf(v); return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r)); }

// computeResizeAreaTab: (dst index, src index, weight) triples, dst-major, as CSR
void area_tab(int ssize, int dsize, double scale, std::vector<int32_t>& start, std::vector<int32_t>& si, std::vector<int32_t>& alpha) {
    for (int dx = 0; dx < dsize; ++dx) {
        start.push_back((int32_t)si.size());
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) { si.push_back(sx1 - 1); alpha.push_back(fbits((float)((sx1 - fsx1) / cell))); }
        for (int sx = sx1; sx < sx2; ++sx) { si.push_back(sx); alpha.push_back(fbits(float(1.0 / cell))); }
        if (fsx2 - sx2 > 1e-3) { si.push_back(sx2); alpha.push_back(fbits((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell))); }
    }
    start.push_back((int32_t)si.size());
}

struct ResizeK {
    const uint8_t* src; uint8_t* dst; const int32_t* tab;
    int n, H, W, C, Ho, Wo, binarize;
};

__global__ __launch_bounds__(256) void resize_u8_kernel(const ResizeK p) {
    const int32_t* t = p.tab;
    const int kind = t[0];
    const int64_t total = (int64_t)p.n * p.Ho * p.Wo;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int dx = (int)(idx % p.Wo), dy = (int)((idx / p.Wo) % p.Ho);
        const int64_t img = idx / ((int64_t)p.Wo * p.Ho);
        const uint8_t* S = p.src + img * p.H * p.W * p.C;
        uint8_t* D = p.dst + (img * p.Ho * p.Wo + (int64_t)dy * p.Wo + dx) * p.C;
        if (kind == RK_COPY) {
            for (int c = 0; c < p.C; ++c) { const uint8_t v = S[((int64_t)dy * p.W + dx) * p.C + c]; D[c] = p.binarize ? (v > 0) : v; }
        } else if (kind == RK_NEAREST) {
            const int sx = t[t[1] + dx], sy = t[t[3] + dy];
            for (int c = 0; c < p.C; ++c) { const uint8_t v = S[((int64_t)sy * p.W + sx) * p.C + c]; D[c] = p.binarize ? (v > 0) : v; }
        } else if (kind == RK_LINEAR) {
            const int sx = t[t[1] + dx], a0 = t[t[2] + 2 * dx], a1 = t[t[2] + 2 * dx + 1], xmax = t[5];
            const int sy0 = t[t[3] + dy], b0 = t[t[4] + 2 * dy], b1 = t[t[4] + 2 * dy + 1];
            const int r0 = min(max(sy0, 0), p.H - 1), r1 = min(max(sy0 + 1, 0), p.H - 1);       // rows clipped at fetch time
            const uint8_t* R0 = S + (int64_t)r0 * p.W * p.C; const uint8_t* R1 = S + (int64_t)r1 * p.W * p.C;
            if (p.binarize) {
                // the float path of a {0, 1} mask followed by `> 0`: any tap with a positive weight over a positive sample
                float fx, fy; int32_t bx = t[t[7] + dx], by = t[t[7] + p.Wo + dy];
                memcpy(&fx, &bx, 4); memcpy(&fy, &by, 4);
                const int sx1 = min(sx + 1, p.W - 1);
                for (int c = 0; c < p.C; ++c) {
                    const bool h0 = R0[sx * p.C + c] > 0 || (fx > 0.f && R0[sx1 * p.C + c] > 0);
                    const bool h1 = R1[sx * p.C + c] > 0 || (fx > 0.f && R1[sx1 * p.C + c] > 0);
                    D[c] = (h0 || (fy > 0.f && h1)) ? 1 : 0;
                }
            } else {
                for (int c = 0; c < p.C; ++c) {
                    int h0, h1;
                    if (dx < xmax) {
                        h0 = R0[sx * p.C + c] * a0 + R0[(sx + 1) * p.C + c] * a1;
                        h1 = R1[sx * p.C + c] * a0 + R1[(sx + 1) * p.C + c] * a1;
                    } else {
                        h0 = R0[sx * p.C + c] * 2048; h1 = R1[sx * p.C + c] * 2048;
                    }
                    D[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
                }
            }
        } else if (kind == RK_AREA) {
            const int32_t* xs = t + t[1]; const int32_t* xa = t + t[2]; const int32_t* ys = t + t[3]; const int32_t* ya = t + t[4];
            const int32_t* xsi = xs + p.Wo + 1; const int32_t* ysi = ys + p.Ho + 1;
            for (int c = 0; c < p.C; ++c) {
                float sum = 0.f;
                for (int j = ys[dy]; j < ys[dy + 1]; ++j) {
                    float beta; memcpy(&beta, &ya[j], 4);
                    const uint8_t* R = S + (int64_t)ysi[j] * p.W * p.C;
                    float buf = 0.f;
                    for (int k = xs[dx]; k < xs[dx + 1]; ++k) {
                        float al; memcpy(&al, &xa[k], 4);
                        buf = __fadd_rn(buf, __fmul_rn((float)R[xsi[k] * p.C + c], al));
                    }
                    sum = (j == ys[dy]) ? __fmul_rn(beta, buf) : __fadd_rn(sum, __fmul_rn(beta, buf));
                }
                int v = __float2int_rn(sum);
                D[c] = p.binarize ? (uint8_t)(sum > 0.f) : (uint8_t)min(max(v, 0), 255);     // positive weights: sum > 0 iff a covered sample is
            }
        } else {    // RK_AREA_FAST
            const int kx = t[5], ky = t[6];
            for (int c = 0; c < p.C; ++c) {
                int sum = 0;
                for (int yy = 0; yy < ky; ++yy)
                    for (int xx = 0; xx < kx; ++xx) sum += S[((int64_t)(dy * ky + yy) * p.W + dx * kx + xx) * p.C + c];
                int v;
                if (kx == 2 && ky == 2) v = (sum + 2) >> 2;
                else v = __float2int_rn(__fmul_rn((float)sum, 1.f / (float)(kx * ky)));
                D[c] = p.binarize ? (uint8_t)(sum > 0) : (uint8_t)min(max(v, 0), 255);
            }
        }
    }
}

}  // namespace

// Host side: the table for one (interpolation, source size, destination size).  Returns the number of int32 words
// (written to `tab` when cap is large enough), < 0 on bad arguments.  Pure host arithmetic: works without a GPU.
extern "C" int64_t pc_resize_tables(int interpolation, int H, int W, int Ho, int Wo, int32_t* tab, int64_t cap) {
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1 || (interpolation != 0 && interpolation != 1 && interpolation != 3)) {
        pc_set_error("pc_resize_tables: interpolation %d (0 nearest, 1 linear, 3 area), %dx%d -> %dx%d", interpolation, H, W, Ho, Wo);
        return PC_E_ARG;
    }
    std::vector<int32_t> out(RHDR, 0);
    const double inv_scale_x = (double)Wo / W, inv_scale_y = (double)Ho / H;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    const bool is_area_fast = std::abs(scale_x - iscale_x) < DBL_EPSILON && std::abs(scale_y - iscale_y) < DBL_EPSILON;
    if (Ho == H && Wo == W) {
        out[0] = RK_COPY;
    } else if (interpolation == 0) {
        out[0] = RK_NEAREST;
        const double ifx = 1. / inv_scale_x, ify = 1. / inv_scale_y;
        out[1] = (int32_t)out.size();
        for (int x = 0; x < Wo; ++x) out.push_back(std::min(cv_floor(x * ifx), W - 1));
        out[3] = (int32_t)out.size();
        for (int y = 0; y < Ho; ++y) out.push_back(std::min(cv_floor(y * ify), H - 1));
    } else {
        if (interpolation == 1 && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = 3;
        if (interpolation == 3 && scale_x >= 1 && scale_y >= 1) {
            if (is_area_fast) {
                out[0] = RK_AREA_FAST; out[5] = iscale_x; out[6] = iscale_y;
            } else {
                out[0] = RK_AREA;
                std::vector<int32_t> st, si, al;
                area_tab(W, Wo, scale_x, st, si, al);
                out[1] = (int32_t)out.size(); out.insert(out.end(), st.begin(), st.end()); out.insert(out.end(), si.begin(), si.end());
                out[2] = (int32_t)out.size(); out.insert(out.end(), al.begin(), al.end());
                out[5] = (int32_t)si.size();
                st.clear(); si.clear(); al.clear();
                area_tab(H, Ho, scale_y, st, si, al);
                out[3] = (int32_t)out.size(); out.insert(out.end(), st.begin(), st.end()); out.insert(out.end(), si.begin(), si.end());
                out[4] = (int32_t)out.size(); out.insert(out.end(), al.begin(), al.end());
                out[6] = (int32_t)si.size();
            }
        } else {
            out[0] = RK_LINEAR;
            const bool area_mode = interpolation == 3;
            std::vector<int32_t> xofs(Wo), ia(2 * Wo), yofs(Ho), ib(2 * Ho), fxs(Wo), fys(Ho);
            int xmax = Wo;
            for (int dx = 0; dx < Wo; ++dx) {
                float fx; int sx;
                if (!area_mode) { fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cv_floor(fx); fx -= sx; }
                else { sx = cv_floor(dx * scale_x); fx = (float)((dx + 1) - (sx + 1) * inv_scale_x); fx = fx <= 0 ? 0.f : fx - cv_floor(fx); }
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx + 1 >= W) { xmax = std::min(xmax, dx); if (sx >= W - 1) { fx = 0; sx = W - 1; } }
                xofs[dx] = sx; fxs[dx] = fbits(fx);
                ia[2 * dx] = sat_short((1.f - fx) * 2048.f); ia[2 * dx + 1] = sat_short(fx * 2048.f);
            }
            for (int dy = 0; dy < Ho; ++dy) {
                float fy; int sy;
                if (!area_mode) { fy = (float)((dy + 0.5) * scale_y - 0.5); sy = cv_floor(fy); fy -= sy; }
                else { sy = cv_floor(dy * scale_y); fy = (float)((dy + 1) - (sy + 1) * inv_scale_y); fy = fy <= 0 ? 0.f : fy - cv_floor(fy); }
                yofs[dy] = sy; fys[dy] = fbits(fy);
                ib[2 * dy] = sat_short((1.f - fy) * 2048.f); ib[2 * dy + 1] = sat_short(fy * 2048.f);
            }
            out[1] = (int32_t)out.size(); out.insert(out.end(), xofs.begin(), xofs.end());
            out[2] = (int32_t)out.size(); out.insert(out.end(), ia.begin(), ia.end());
            out[3] = (int32_t)out.size(); out.insert(out.end(), yofs.begin(), yofs.end());
            out[4] = (int32_t)out.size(); out.insert(out.end(), ib.begin(), ib.end());
            out[5] = xmax;
            out[7] = (int32_t)out.size(); out.insert(out.end(), fxs.begin(), fxs.end()); out.insert(out.end(), fys.begin(), fys.end());
        }
    }
    if (tab && cap >= (int64_t)out.size()) memcpy(tab, out.data(), out.size() * sizeof(int32_t));
    return (int64_t)out.size();
}

extern "C" int pc_resize_u8(const uint8_t* src, int n, int H, int W, int C, int Ho, int Wo, const int32_t* dev_tab, int binarize,
                            uint8_t* dst, pc_stream s) {
    PC_CHECK_ARG(src && dst && dev_tab, "pc_resize_u8: null pointer");
    PC_CHECK_ARG(n >= 1 && H >= 1 && W >= 1 && C >= 1 && C <= 4 && Ho >= 1 && Wo >= 1, "pc_resize_u8: n=%d %dx%dx%d -> %dx%d", n, H, W, C, Ho, Wo);
    ResizeK k; k.src = src; k.dst = dst; k.tab = dev_tab; k.n = n; k.H = H; k.W = W; k.C = C; k.Ho = Ho; k.Wo = Wo; k.binarize = binarize;
    const int64_t total = (int64_t)n * Ho * Wo;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, k);
    PC_CHECK_LAUNCH("resize_u8");
    return PC_OK;
}
