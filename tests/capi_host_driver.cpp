// Host-side walk through libpicons' C-ABI without a GPU, linked against the AddressSanitizer + UBSan build
// (`make -C pi-consistency-activity-detection_amd/csrc asan`): every call below must return through the library's own
// argument checking (PC_E_ARG + pc_last_error) or is pure host arithmetic (workspace sizes, the cv2.resize tables), so any
// out-of-bounds access, leak or undefined behaviour on the host side of the boundary ends the process with a sanitizer report.
// tests/test_capi_cpu.py builds and runs it.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "picons.h"

static int fails = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) { ++fails; std::printf("FAILED %s:%d  %s  [%s]\n", __FILE__, __LINE__, #cond, pc_last_error()); } \
    } while (0)

int main() {
    EXPECT(pc_version() == PC_VERSION);
    EXPECT(pc_last_error() != nullptr);
    char dummy[256] = {0};
    float* fp = reinterpret_cast<float*>(dummy);

    // conv descriptor checks (no launch happens before them)
    pc_conv_desc cd;
    std::memset(&cd, 0, sizeof(cd));
    cd.N = 1; cd.Ti = 1; cd.Hi = 2; cd.Wi = 2; cd.Ci = 3; cd.ldi = 3; cd.Tq = 1; cd.Hq = 2; cd.Wq = 2; cd.To = 1; cd.Ho = 2; cd.Wo = 2; cd.Co = 4; cd.ldo = 4;
    for (int i = 0; i < 3; ++i) { cd.ostr[i] = 1; cd.istr[i] = 1; cd.ntap[i] = 1; cd.istep[i] = 1; cd.wkstep[i] = 1; }
    cd.KT = cd.KH = cd.KW = 1; cd.ldw = 3; cd.groups = 1;
    EXPECT(pc_conv_fwd(&cd, fp, fp, nullptr, nullptr, fp, nullptr, nullptr) == PC_E_ARG);
    EXPECT(std::strstr(pc_last_error(), "multiples of 4") != nullptr);
    EXPECT(pc_conv_fwd(nullptr, fp, fp, nullptr, nullptr, fp, nullptr, nullptr) == PC_E_ARG);
    cd.Ci = cd.ldi = cd.ldw = 4;
    EXPECT(pc_conv_fwd(&cd, nullptr, fp, nullptr, nullptr, fp, nullptr, nullptr) == PC_E_ARG);
    EXPECT(pc_conv_bnpart_rows(&cd) >= 1);
    pc_wgrad_desc wd;
    std::memset(&wd, 0, sizeof(wd));
    EXPECT(pc_conv_wgrad(&wd, nullptr, nullptr, nullptr, nullptr) == PC_E_ARG);
    EXPECT(pc_conv_wgrad(nullptr, fp, fp, fp, nullptr) == PC_E_ARG);
    EXPECT(pc_wgrad_slices(nullptr) == -1 && pc_wgrad_uses_x6(nullptr) == 0);
    {   // round 6: K-slice counts of every route (host arithmetic only: the split of the K range, the stem's per-tap slice table), the family a
        // problem's launch belongs to, and the checks in front of a launch that would write more slice images than the workspace holds
        struct { int N, T, H, W, Cd, Cs, k, s, flags; } cases[] = {
            {16, 4, 112, 112, 64, 4, 7, 2, PC_WG_CS3}, {16, 4, 112, 112, 64, 4, 7, 2, PC_WG_CS3 | PC_WG_X6},          // stem: wgrad4 / wgrad4_x6
            {16, 4, 112, 112, 64, 64, 3, 1, PC_WG_X6}, {16, 2, 28, 28, 128, 96, 3, 1, 0},                              // row-segment
            {16, 1, 28, 28, 288, 512, 1, 1, PC_WG_X6}, {16, 2, 56, 56, 64, 64, 1, 1, PC_WG_X6}, {2, 1, 20, 20, 136, 48, 3, 1, 0}};   // generic
        for (const auto& c : cases) {
            pc_wgrad_desc w;
            std::memset(&w, 0, sizeof(w));
            w.N = c.N; w.Tq = c.T; w.Hq = c.H; w.Wq = c.W; w.Cd = c.Cd; w.ldd = c.Cd;
            w.Ts = c.T * c.s; w.Hs = c.H * c.s; w.Ws = c.W * c.s; w.Cs = c.Cs; w.lds = c.Cs;
            for (int i = 0; i < 3; ++i) { w.istr[i] = c.s; w.ntap[i] = c.k; w.ioff0[i] = -(c.k / 2 - (c.s == 2 ? 1 : 0)); w.istep[i] = 1; }
            w.KT = w.KH = w.KW = c.k; w.flags = c.flags;
            const int ns = pc_wgrad_slices(&w);
            EXPECT(ns >= 1 && ns <= 1024);
            EXPECT(pc_wgrad_uses_x6(&w) == ((c.flags & PC_WG_X6) ? 1 : 0));
            double wk[5];
            EXPECT(pc_wgrad_work(&w, 0, c.Cs == 4 ? 3 : 0, wk) == PC_OK && wk[0] >= wk[1] && wk[1] > 0);
            if (ns > 1) {
                w.ws_slices = ns - 1;                                                   // one image short: refused before any launch
                EXPECT(pc_conv_wgrad(&w, fp, fp, fp, nullptr) == PC_E_ARG);
                EXPECT(std::strstr(pc_last_error(), "slice images") != nullptr);
            }
            w.ws_slices = ns; w.splitk = -1;                                            // plain stores and slice images exclude each other
            EXPECT(pc_conv_wgrad(&w, fp, fp, fp, nullptr) == PC_E_ARG);
        }
        EXPECT(pc_wgrad_fold_group() >= 2);
        EXPECT(pc_wgrad_fold(nullptr, 64, 4, nullptr) == PC_E_ARG && pc_wgrad_fold(fp, 6, 4, nullptr) == PC_E_ARG);
        pc_transpose_job tj;
        std::memset(&tj, 0, sizeof(tj));
        tj.src = (uint64_t)(uintptr_t)fp; tj.dst = (uint64_t)(uintptr_t)fp; tj.batch = 1; tj.R = 4; tj.C = 3; tj.src_ld = 3; tj.dst_ld = 4;
        tj.nslices = 2; tj.slice_stride = 0;                                            // summed images need a stride ...
        EXPECT(pc_transpose_multi(&tj, 1, nullptr) == PC_E_ARG);
        tj.slice_stride = 64;                                                           // ... and 16-byte rows (src_ld % 4)
        EXPECT(pc_transpose_multi(&tj, 1, nullptr) == PC_E_ARG);
    }

    // the merged tail's ordered reductions: workspace sizing and argument checks (host arithmetic, refused before any launch)
    EXPECT(pc_tail6_bias_sums_ws_floats(16, 4, 112, 112) == 16 * 98 * 32 && pc_tail6_bias_sums_ws_floats(0, 4, 112, 112) == -1);
    EXPECT(pc_tail6_bias_sums_ws_floats(2, 1, 3, 3) == 2 * 32);                        // small frames: one block per clip-pass
    {
        int32_t ns8[8] = {4, 1, 1, 1, 2, 1, 1, -1};
        EXPECT(pc_tail6_wgrad_map_slices(fp, nullptr, 2, 8, fp, nullptr) == PC_E_ARG);
        EXPECT(pc_tail6_wgrad_map_slices(fp, ns8, 2, 8, fp, nullptr) == PC_E_ARG);      // a negative slice count
        EXPECT(pc_tail6_bias_sums_ws(nullptr, 2, 1, 3, 3, fp, fp, nullptr) == PC_E_ARG);
        EXPECT(pc_tail_grads_ws_floats(16, 128, 128) == 16 * 16 * 128 * 32 && pc_tail_grads_ws_floats(16, 12, 128) == -1);
    }

    // workspace sizing: host arithmetic
    EXPECT(pc_bn_bwd_ws_floats(802816, 64, 2) > 0);
    EXPECT(pc_act_bwd_ws_floats(802816, 64) > 0);
    EXPECT(pc_em_ws_floats(6400, 32, 24) > 0);
    pc_loss_desc ld;
    std::memset(&ld, 0, sizeof(ld));
    ld.B = 8; ld.T = 8; ld.H = 224; ld.W = 224;
    EXPECT(pc_loss_ws_floats(&ld) > 0);

    // the op-list runner
    EXPECT(pc_run_ops(nullptr, 3, nullptr) == PC_E_ARG);
    EXPECT(pc_run_ops_lanes(nullptr, 0, nullptr, 0) == PC_E_ARG);
    pc_stream lanes[2] = {nullptr, nullptr};
    EXPECT(pc_run_ops_lanes(nullptr, 0, lanes, 99) == PC_E_ARG);
    std::vector<pc_op> ops(2);
    std::memset(ops.data(), 0, ops.size() * sizeof(pc_op));
    ops[0].kind = 9999;
    EXPECT(pc_run_ops(ops.data(), 1, nullptr) == PC_E_ARG);
    EXPECT(std::strstr(pc_last_error(), "unknown op kind") != nullptr);
    float ms = 0.f; int cnt = 0;
    EXPECT(pc_run_ops_timed(ops.data(), 1, 0, &ms, &cnt, lanes, 1) == PC_E_ARG);
    EXPECT(pc_run_ops_timed_collect(&ms, &cnt) == PC_OK && cnt == 0);

    // input pipeline
    int32_t span[8] = {0, 1, 2, 3, 4, 5, 6, 99};
    const uint8_t* u8 = reinterpret_cast<const uint8_t*>(dummy);
    EXPECT(pc_clip_from_u8(nullptr, 8, 240, 320, span, 0, 0, 224, nullptr, 0, fp, fp, fp, nullptr) == PC_E_ARG);
    EXPECT(pc_clip_from_u8(u8, 8, 240, 320, span, 100, 0, 224, nullptr, 0, fp, fp, fp, nullptr) == PC_E_ARG);      // crop outside
    EXPECT(pc_clip_from_u8(u8, 8, 240, 320, span, 0, 0, 224, nullptr, 0, fp, fp, fp, nullptr) == PC_E_ARG);        // frame 99 of 8
    EXPECT(pc_clip_from_u8_masks(u8, 8, 240, 320, span, 0, 0, 224, nullptr, nullptr, fp, fp, fp, nullptr, nullptr) == PC_E_ARG);
    EXPECT(pc_resize_u8(nullptr, 1, 240, 320, 3, 256, 256, nullptr, 0, nullptr, nullptr) == PC_E_ARG);
    EXPECT(pc_resize_u8(u8, 1, 240, 320, 9, 256, 256, reinterpret_cast<const int32_t*>(dummy), 0, reinterpret_cast<uint8_t*>(dummy), nullptr) == PC_E_ARG);
    EXPECT(pc_resize_tables(2, 240, 320, 256, 256, nullptr, 0) == PC_E_ARG);
    EXPECT(pc_resize_tables(1, 240, 0, 256, 256, nullptr, 0) == PC_E_ARG);
    // cv2.resize tables: host arithmetic into caller memory, sized by the first call; the guard words must survive
    const int cases[][5] = {{3, 240, 320, 256, 256}, {0, 240, 320, 256, 256}, {1, 224, 224, 112, 112}, {1, 224, 224, 160, 160}, {1, 224, 224, 300, 256},
                            {3, 240, 320, 96, 128}, {3, 240, 320, 80, 80}, {3, 250, 333, 100, 111}, {3, 64, 64, 64, 64}, {0, 37, 53, 90, 17},
                            {3, 1, 1, 7, 5}, {1, 2, 3, 1, 1}, {3, 1000, 3, 7, 2}};
    for (const auto& c : cases) {
        const int64_t n = pc_resize_tables(c[0], c[1], c[2], c[3], c[4], nullptr, 0);
        EXPECT(n >= 8);
        std::vector<int32_t> tab((size_t)n + 2, 0x5a5a5a5a);
        EXPECT(pc_resize_tables(c[0], c[1], c[2], c[3], c[4], tab.data() + 1, n) == n);
        EXPECT(tab[0] == 0x5a5a5a5a && tab[(size_t)n + 1] == 0x5a5a5a5a);
        EXPECT(tab[1] >= 0 && tab[1] <= 4);
        std::vector<int32_t> small((size_t)n - 1, 7);
        EXPECT(pc_resize_tables(c[0], c[1], c[2], c[3], c[4], small.data(), n - 1) == n && small[0] == 7);          // too small: untouched
    }

    // evaluation metrics / misc entry points with null pointers
    EXPECT(pc_seg_frame_counts(nullptr, nullptr, 1, 50176, nullptr, nullptr) == PC_E_ARG);
    EXPECT(pc_transpose_multi(nullptr, 3, nullptr) == PC_E_ARG);

    // Winograd kernels, both forms: block / pitch choice and work accounting are host arithmetic (choose_block, choose_pitch: the loops over lane
    // groups and candidate pitches run here under the sanitizers for every frame size), argument checks stop in front of any launch
    {
        double w3[3];
        for (int m = 2; m <= 4; m += 2)
            for (int H = 4; H <= 256; H += 4)
                for (int W = 4; W <= 256; W += (W < 64 ? 4 : 36)) {
                    for (int N = 2; N <= 4; N += 2) {          // N = 4: F(2x2, 3x3) strip mode where the tile grid is a multiple of 14 wide (plane pairs)
                        pc_wino_desc d{};
                        d.N = N; d.T = 3; d.H = H; d.W = W; d.Ci = 24; d.ldi = 24; d.Co = 72; d.ldo = 72; d.KT = 3; d.Ti = 3; d.ta = 1; d.tc = -1; d.tden = 1; d.m = m;
                        d.flags = (m == 2 && N == 4) ? PC_F_STRIPS : 0;
                        EXPECT(pc_wino_work(&d, w3) == PC_OK && w3[0] >= w3[1] && w3[1] > 0 && w3[2] > 0);
                        EXPECT(pc_wino_bnpart_rows(&d) > 0);
                        EXPECT(pc_wino_conv(&d, nullptr, fp, nullptr, fp, nullptr, nullptr) == PC_E_ARG);
                        if (m == 2 && N == 4 && H == W && (H == 28 || H == 56))       // strips: H / 4 blocks per plane pair and 28-wide column block, two partial rows each
                            EXPECT(pc_wino_bnpart_rows(&d) == (N / 2) * d.T * (H / 4) * (W / 28) * 2);
                        if (m == 2 && H == 112 && W == 112)                           // 56 x 56 tiles are exactly 49 rectangles of 8 x 8: strips would be MORE blocks
                            EXPECT(pc_wino_bnpart_rows(&d) == N * d.T * 49 * 2);
                    }
                }
        pc_wino_desc d{};
        d.N = 1; d.T = 1; d.H = 14; d.W = 14; d.Ci = 8; d.ldi = 8; d.Co = 8; d.ldo = 8; d.KT = 3; d.Ti = 1; d.ta = 1; d.tc = -1; d.tden = 1;
        d.m = 4;
        EXPECT(pc_wino_work(&d, w3) == PC_E_ARG && std::strstr(pc_last_error(), "multiples of 4") != nullptr && pc_wino_bnpart_rows(&d) == -1);
        d.m = 3;
        EXPECT(pc_wino_work(&d, w3) == PC_E_ARG && std::strstr(pc_last_error(), "m must be") != nullptr && pc_wino_bnpart_rows(&d) == -1);
        d.m = 2; d.H = 15;
        EXPECT(pc_wino_work(&d, w3) == PC_E_ARG && pc_wino_work(nullptr, w3) == PC_E_ARG);
        EXPECT(pc_wino_u_floats(64, 64, 3) == 3 * 8 * 8192 && pc_wino4_u_floats(64, 64, 3) == 3 * 16 * 9216);
        EXPECT(pc_wino_u_floats(64, 12, 3) == -1 && pc_wino4_u_floats(64, 12, 3) == -1 && pc_wino4_u_floats(64, 64, 2) == -1);
        EXPECT(pc_wino_weights(nullptr, 1, 1, 1, 64, 64, 3, 0, fp, nullptr) == PC_E_ARG && pc_wino4_weights(fp, 1, 1, 1, 64, 64, 3, 0, nullptr, nullptr) == PC_E_ARG);
        EXPECT(pc_wino4_weights(fp, 1, 1, 1, 64, 12, 3, 0, fp, nullptr) == PC_E_ARG);
    }
    if (fails) { std::printf("%d host-side checks failed\n", fails); return 1; }
    std::printf("capi host driver: all checks passed\n");
    return 0;
}
