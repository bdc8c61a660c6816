// Input pipeline on the device: what /root/reference/datasets/ucf_dataloader.py does per sample on
// the host in numpy after decoding -- pick 8 frames, crop 224x224, /255, the horizontally flipped copy, and the
// foreground mask from the per-frame boxes (:146-173, :204-221) -- from the uint8 frames already in HBM, writing the fp32
// NCDHW staging tensors the step engine consumes.  HBM-bound: 3 bytes read, 28 bytes written per pixel.
#include "common.h"

namespace {

struct ClipK {
    const uint8_t* video; int F, H, W;
    int span[8]; int h0, w0, S;          // frame ids, crop origin, crop size (224)
    const int32_t* rects; int R;         // [8][R][4] = x0, x1, y0, y1 in frame coordinates (empty: x1 <= x0)
    const uint8_t* maskf; int valid[8]; float* mask_cls;   // JHMDB form: per-pixel truth frames [F][H][W] (> 0 = foreground), frames that carry truth
    float* data; float* aug; float* mask;   // [3][8][S][S], [3][8][S][S], [8][S][S]
};

__global__ __launch_bounds__(256) void clip_from_u8_kernel(const ClipK p) {
    const int S = p.S;
    const int64_t total = (int64_t)8 * S * S;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int w = (int)(idx % S), h = (int)((idx / S) % S), t = (int)(idx / ((int64_t)S * S));
        const int f = p.span[t], y = h + p.h0, x = w + p.w0;
        const uint8_t* px = p.video + (((size_t)f * p.H + y) * p.W + x) * 3;
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
                       const uint8_t* maskf, const int32_t* valid8, float* data, float* aug, float* mask, float* mask_cls, pc_stream s);

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

static int clip_launch(const uint8_t* video, int F, int H, int W, const int32_t* span8, int h0, int w0, int S, const int32_t* rects, int R,
                       const uint8_t* maskf, const int32_t* valid8, float* data, float* aug, float* mask, float* mask_cls, pc_stream s) {
    PC_CHECK_ARG(video && span8 && data && aug && mask && (rects || R == 0), "pc_clip_from_u8: null pointer");
    PC_CHECK_ARG(F >= 1 && S >= 1 && h0 >= 0 && w0 >= 0 && h0 + S <= H && w0 + S <= W && R >= 0, "pc_clip_from_u8: crop %d+%d x %d+%d outside %d x %d", h0, S, w0, S, H, W);
    ClipK k;
    k.video = video; k.F = F; k.H = H; k.W = W; k.h0 = h0; k.w0 = w0; k.S = S; k.rects = rects; k.R = R; k.data = data; k.aug = aug; k.mask = mask;
    k.maskf = maskf; k.mask_cls = mask_cls;
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
