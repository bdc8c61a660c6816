// Evaluation metrics on the device: the per-frame intersection / union counts of the thresholded masks
// against the truth and the per-class f-mAP / v-mAP hit tables the reference accumulates on the host in numpy
// (/root/reference/evaluate_ucf101.py:142-183).  HBM-bound: every logit and every truth pixel is read once.
#include "common.h"

namespace {

constexpr int NTHR = 20;          // evaluate_ucf101.py:72: thresholds k / 20, k = 0..19

// counts[frame] = {#(pred + gt == 2), #(pred + gt != 0), #(gt != 0)} with pred = sigmoid(logit) >= 0.5 (evaluate_ucf101.py:128,148-149,160-161)
__global__ __launch_bounds__(256) void seg_frame_counts_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                               int64_t pix4, int blocks_per_frame, int32_t* __restrict__ counts) {
    const int frame = blockIdx.x / blocks_per_frame, part = blockIdx.x % blocks_per_frame;
    const f32x4* lp = (const f32x4*)(logits) + (size_t)frame * pix4;
    const f32x4* gp = (const f32x4*)(gt) + (size_t)frame * pix4;
    int inter = 0, uni = 0, gnz = 0;
    for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < pix4; i += (int64_t)blocks_per_frame * 256) {
        const f32x4 x = lp[i], g = gp[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // sigmoid(x) >= 0.5 in fp32: certainly true for x >= 0 (exp(-x) <= 1 => 1 + e <= 2, division is monotone) and
            // certainly false below -1e-6 (1 + e >= 2.000001 > 2); only in between does the rounding of exp / the sum decide
            const float xe = x[e];
            const bool on = xe >= 0.f ? true : (xe < -1e-6f ? false : (1.0f / (1.0f + expf(-xe))) >= 0.5f);
            const float v = (on ? 1.0f : 0.0f) + g[e];
            inter += v == 2.0f;
            uni += v != 0.0f;
            gnz += g[e] != 0.0f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inter += __shfl_xor(inter, o, 64); uni += __shfl_xor(uni, o, 64); gnz += __shfl_xor(gnz, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (inter) atomicAdd(counts + frame * 3 + 0, inter);
        if (uni) atomicAdd(counts + frame * 3 + 1, uni);
        if (gnz) atomicAdd(counts + frame * 3 + 2, gnz);
    }
}

// One video (evaluate_ucf101.py:153-183): frames with truth count towards n_frames[label] and, per threshold k with
// inter/union >= k/20 (float64 ratio against the float32 threshold, as numpy compares them), towards frame_hits[label][k];
// the video's summed inter / union does the same for video_hits; n_vids[label] += 1.
__global__ void map_accumulate_kernel(const int32_t* __restrict__ counts, int64_t nframes, int label, int32_t* frame_hits,
                                      int32_t* video_hits, int32_t* n_frames, int32_t* n_vids) {
    const int k = threadIdx.x;
    if (k >= NTHR) return;
    const double thr = (double)((float)k / 20.0f);
    long long vi = 0, vu = 0;
    int hits = 0, nf = 0;
    for (int64_t f = 0; f < nframes; ++f) {
        const int inter = counts[f * 3], uni = counts[f * 3 + 1], gnz = counts[f * 3 + 2];
        if (gnz == 0) continue;
        ++nf; vi += inter; vu += uni;
        hits += ((double)inter / (double)uni) >= thr;
    }
    frame_hits[label * NTHR + k] += hits;
    if (vu > 0) video_hits[label * NTHR + k] += ((double)vi / (double)vu) >= thr;
    if (k == 0) { n_frames[label] += nf; n_vids[label] += 1; }
}

}  // namespace

extern "C" int pc_seg_frame_counts(const float* logits, const float* gt, int64_t nframes, int64_t pix, int32_t* counts, pc_stream s_) {
    hipStream_t s = (hipStream_t)s_;
    PC_CHECK_ARG(logits && gt && counts && nframes >= 1 && pix >= 4 && pix % 4 == 0, "pc_seg_frame_counts: bad args (frames=%lld pix=%lld)",
                 (long long)nframes, (long long)pix);
    PC_CHECK_ARG(((uintptr_t)logits % 16 == 0) && ((uintptr_t)gt % 16 == 0), "pc_seg_frame_counts: 16-byte alignment");
    PC_CHECK_ARG(pix < (1ll << 31) && nframes < (1ll << 24), "pc_seg_frame_counts: size out of range");
    if (hipMemsetAsync(counts, 0, (size_t)nframes * 3 * sizeof(int32_t), s) != hipSuccess) PC_CHECK_ARG(false, "pc_seg_frame_counts: memset failed");
    const int64_t pix4 = pix / 4;
    int bpf = (int)((pix4 + 256 * 4 - 1) / (256 * 4));        // ~4 float4 per thread
    if (bpf < 1) bpf = 1;
    if (bpf > 64) bpf = 64;
    hipLaunchKernelGGL(seg_frame_counts_kernel, dim3((unsigned)(nframes * bpf)), dim3(256), 0, s, logits, gt, pix4, bpf, counts);
    PC_CHECK_LAUNCH("seg_frame_counts");
    return PC_OK;
}

extern "C" int pc_map_accumulate(const int32_t* counts, int64_t nframes, int label, int ncls, int32_t* frame_hits, int32_t* video_hits,
                                 int32_t* n_frames, int32_t* n_vids, pc_stream s_) {
    PC_CHECK_ARG(counts && frame_hits && video_hits && n_frames && n_vids && nframes >= 1, "pc_map_accumulate: null / empty");
    PC_CHECK_ARG(label >= 0 && label < ncls, "pc_map_accumulate: label %d outside [0, %d)", label, ncls);
    hipLaunchKernelGGL(map_accumulate_kernel, dim3(1), dim3(64), 0, (hipStream_t)s_, counts, nframes, label, frame_hits, video_hits, n_frames, n_vids);
    PC_CHECK_LAUNCH("map_accumulate");
    return PC_OK;
}
