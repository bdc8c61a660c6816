// libpicons.so runtime glue: error reporting, version, and the op-list runner that replays a
// host-built plan (pi-consistency-activity-detection_amd/plan.py) with no per-op host round trip.
#include "common.h"
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

void pc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pc_last_error(void) { return g_err; }
extern "C" int pc_version(void) { return PC_VERSION; }

extern "C" int pc_transpose_batched(const float*, int, int, int, int64_t, int, float*, int64_t, int, int, pc_stream);

#define P(T, k) ((T)(uintptr_t)op.p[k])

static int run_one(const pc_op& op, pc_stream s) {
    switch (op.kind) {
        case PC_OP_CONV: {
            pc_conv_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_conv_fwd(&d, P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const float*, 3), P(float*, 4), P(float*, 5), s);
        }
        case PC_OP_CONV_X6: {
            pc_conv_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_conv_fwd_x6_ws(&d, P(const float*, 0), P(const uint16_t*, 1), op.l[0], P(const float*, 2), P(const float*, 3), P(float*, 4), P(float*, 5), P(float*, 6), op.l[1], s);
        }
        case PC_OP_SPLIT_PLANES:
            return pc_split_planes(P(const float*, 0), P(uint16_t*, 1), op.l[0], op.l[1], s);
        case PC_OP_WSPEC_MASTER_PLANES:
            return pc_wspec_master_planes(P(const float*, 0), P(const float*, 1), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7],
                                          P(uint16_t*, 2), P(uint16_t*, 3), op.l[0], s);
        case PC_OP_SPLIT_PLANES_MULTI:
            return pc_split_planes_multi(P(const pc_split_job*, 0), op.i[0], s);
        case PC_OP_WGRAD: {
            pc_wgrad_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_conv_wgrad(&d, P(const float*, 0), P(const float*, 1), P(float*, 2), s);
        }
        case PC_OP_BN_FINALIZE:
            return pc_bn_finalize_ws(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.l[0], P(const float*, 1), P(const float*, 2), op.f[0], op.f[1],
                                     P(float*, 3), P(float*, 4), P(float*, 5), P(float*, 6), s);
        case PC_OP_BN_APPLY:
            return pc_bn_apply(P(const float*, 0), op.i[0], P(const float*, 1), op.i[1], op.l[0], op.i[2], P(float*, 2), op.i[3], op.i[4], s);
        case PC_OP_BN_FIN_APPLY:
            return pc_bn_finalize_apply(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.l[0], P(const float*, 1), P(const float*, 2), op.f[0], op.f[1], P(float*, 3),
                                        P(float*, 4), P(float*, 5), P(const float*, 6), op.i[3], op.l[1], P(float*, 7), op.i[4], op.i[5], s);
        case PC_OP_BN_EVAL_STAT:
            return pc_bn_eval_stat(P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const float*, 3), op.f[0], op.i[0], P(float*, 4), s);
        case PC_OP_BN_BWD:
            return pc_bn_bwd(P(const float*, 0), op.i[0], P(const float*, 1), op.i[1], P(const float*, 2), op.i[2], op.l[0], op.i[3], op.i[4],
                             P(float*, 3), op.i[5], P(float*, 4), P(float*, 5), op.i[6], P(float*, 6), s);
        case PC_OP_POOL_FWD: {
            pc_pool_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_maxpool_fwd(&d, P(const float*, 0), P(float*, 1), P(uint8_t*, 2), s);
        }
        case PC_OP_POOL_BWD: {
            pc_pool_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_maxpool_bwd(&d, P(const float*, 0), P(const uint8_t*, 1), P(float*, 2), op.i[19], s);
        }
        case PC_OP_CHSCALE:
            return pc_channel_scale(P(const float*, 0), op.i[0], P(const float*, 1), op.i[1], op.l[0], op.i[2], P(float*, 2), op.i[3], op.i[4], s);
        case PC_OP_ACT_BWD:
            return pc_act_bwd(P(const float*, 0), op.i[0], P(const float*, 1), op.i[1], op.i[2], op.i[3], op.l[0], P(float*, 2), op.i[4],
                              P(float*, 3), op.i[5], P(float*, 4), s);
        case PC_OP_TO_NDHWC:
            if (op.i[1] == 0) return PC_OK;        // N = 0: the clip is already where the network reads it (StepEngine.sample_stager re-points its readers)
            return pc_ncdhw_to_ndhwc(P(const void*, 0), op.i[0], op.i[1], op.i[2], op.l[0], op.i[3], op.i[4], op.i[5], P(float*, 1), s);
        case PC_OP_TO_NCDHW:
            return pc_ndhwc_to_ncdhw(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.l[0], P(float*, 1), s);
        case PC_OP_TRANSPOSE:
            if (op.i[6] > 1) {          // the source is i[6] K-slice images of a weight gradient, l[2] floats apart: added in slice order on the way
                pc_transpose_job j;
                memset(&j, 0, sizeof(j));
                j.src = op.p[0]; j.dst = op.p[1]; j.src_batch_stride = op.l[0]; j.dst_batch_stride = op.l[1];
                j.batch = op.i[0]; j.R = op.i[1]; j.C = op.i[2]; j.src_ld = op.i[3]; j.dst_ld = op.i[4]; j.accum = op.i[5];
                j.nslices = op.i[6]; j.slice_stride = op.l[2];
                return pc_transpose_multi(&j, 1, s);
            }
            return pc_transpose_batched(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.l[0], op.i[3], P(float*, 1), op.l[1], op.i[4], op.i[5], s);
        case PC_OP_FILL:
            return pc_fill(P(float*, 0), op.l[0], op.f[0], s);
        case PC_OP_AXPY:
            return pc_axpy(P(float*, 0), P(const float*, 1), op.l[0], op.f[0], s);
        case PC_OP_EM_FWD:
            return pc_em_routing_fwd(P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const float*, 3), op.i[0], op.i[1], op.i[2], P(float*, 4), P(float*, 5), s);
        case PC_OP_EM_BWD:
            return pc_em_routing_bwd(P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const float*, 3), P(const float*, 4), op.i[0], op.i[1],
                                     op.i[2], P(float*, 5), P(float*, 6), P(float*, 7), P(float*, 8), P(float*, 9), P(const float*, 10), s);
        case PC_OP_CMASK_FWD:
            return pc_class_mask_fwd(P(const float*, 0), op.i[0], op.i[1], op.i[2], P(const float*, 1), P(const int32_t*, 2), op.i[3], P(float*, 3),
                                     P(float*, 4), P(float*, 5), s);
        case PC_OP_CMASK_BWD:
            return pc_class_mask_bwd(P(const float*, 0), P(const float*, 1), P(const float*, 2), op.i[0], op.i[1], op.i[2], P(float*, 3), s);
        case PC_OP_TAPSUM_FWD:
            return pc_tapsum_fwd(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.i[3], P(const float*, 1), P(float*, 2), s);
        case PC_OP_TAPSUM_BWD:
            return pc_tapsum_bwd(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.i[3], P(float*, 1), s);
        case PC_OP_LOSS: {
            pc_loss_desc d;
            d.B = op.i[0]; d.T = op.i[1]; d.H = op.i[2]; d.W = op.i[3]; d.bv = op.i[4]; d.gv = op.i[5]; d.n_frames = op.i[6];
            d.predict_maps = op.i[7]; d.jhmdb = op.i[8];
            d.lower_thresh = op.f[0]; d.upper_thresh = op.f[1]; d.bv_wt = op.f[2]; d.gv_wt = op.f[3]; d.wt_loc = op.f[4];
            d.wt_cons = op.f[5]; d.wt_ramp = op.f[6];
            return pc_consistency_loss(&d, P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const int32_t*, 3), P(float*, 4), P(float*, 5),
                                       P(float*, 6), P(float*, 7), P(float*, 8), P(float*, 9), s);
        }
        case PC_OP_SPREAD:
            return pc_spread_loss(P(const float*, 0), P(const float*, 1), P(const int32_t*, 2), op.i[0], op.i[1], op.f[0], op.f[1], P(float*, 3), P(float*, 4), s);
        case PC_OP_ADAM:
            return pc_adam_step(P(float*, 0), P(const float*, 1), P(float*, 2), P(float*, 3), op.l[0], op.f[0], op.f[1], op.f[2], op.f[3], op.i[0], op.f[4], s);
        case PC_OP_TAIL_COMBINE:
            return pc_tail_combine(P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const float*, 3), op.i[0], op.i[1], op.i[2], op.i[3],
                                   op.i[4], P(float*, 4), P(float*, 5), P(float*, 6), s);
        case PC_OP_TAIL_COLSUM:
            return pc_tail_colsum(P(const float*, 0), op.i[0], op.l[0], P(float*, 1), s);
        case PC_OP_TAIL_GRADS:
            return pc_tail_grads_ws(P(const float*, 0), P(const float*, 1), P(const float*, 2), P(const float*, 3), P(const float*, 4), P(const float*, 5),
                                    op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], P(float*, 6), P(float*, 7), P(float*, 8), P(float*, 9), op.i[6],
                                    P(float*, 10), s);
        case PC_OP_COL2IM:
            return pc_col2im(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], P(float*, 1), op.i[6], op.i[7], s);
        case PC_OP_AXIS: {
            pc_axis_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_axis_linear(&d, P(const float*, 0), P(const float*, 1), P(const float*, 2), P(float*, 3), s);
        }
        case PC_OP_WSPEC_FWD:
            return pc_wspec_fwd(P(const float*, 0), P(const float*, 1), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], P(float*, 2), s);
        case PC_OP_WSPEC_BWD:
            return pc_wspec_bwd(P(const float*, 0), P(const float*, 1), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], P(float*, 2), s);
        case PC_OP_WSPEC_MASTER_FWD:
            return pc_wspec_master_fwd(P(const float*, 0), P(const float*, 1), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7],
                                       P(float*, 2), P(float*, 3), s);
        case PC_OP_WSPEC_MASTER_BWD:
            return pc_wspec_master_bwd(P(const float*, 0), P(const float*, 1), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7],
                                       P(float*, 2), op.i[8], s);
        case PC_OP_TAIL6_WEIGHTS:
            return pc_tail6_weights(P(const float*, 0), op.i[0], op.i[1], P(float*, 1), P(float*, 2), s);
        case PC_OP_TAIL6_GATHER:
            return pc_tail6_gather(P(const float*, 0), P(const float*, 1), P(const float*, 2), op.i[0], op.i[1], op.i[2], op.i[3], P(float*, 3), s);
        case PC_OP_TAIL6_SCATTER:
            return pc_tail6_scatter(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.i[3], P(float*, 1), s);
        case PC_OP_TAIL6_WGRAD_MAP:
            return op.i[2] ? pc_tail6_wgrad_map_slices(P(float*, 0), &op.i[3], op.i[0], op.i[1], P(float*, 1), s)
                           : pc_tail6_wgrad_map(P(const float*, 0), op.i[0], op.i[1], P(float*, 1), s);
        case PC_OP_TAIL6_BIAS_SUMS:
            return pc_tail6_bias_sums_ws(P(const float*, 0), op.i[0], op.i[1], op.i[2], op.i[3], P(float*, 1), P(float*, 2), s);
        case PC_OP_TRANSPOSE_MULTI:
            return pc_transpose_multi(P(const pc_transpose_job*, 0), op.i[0], s);
        case PC_OP_WGRAD_FOLD:
            return pc_wgrad_fold(P(float*, 0), op.l[0], op.i[0], s);
        case PC_OP_WGRAD_MULTI:
            return pc_conv_wgrad_multi(P(const pc_wgrad_job*, 0), op.i[0], s);
        case PC_OP_WINO_CONV: {
            pc_wino_desc d;
            memcpy(&d, op.i, sizeof(d));
            return pc_wino_conv(&d, P(const float*, 0), P(const float*, 1), P(const float*, 2), P(float*, 3), P(float*, 4), s);
        }
        case PC_OP_WINO_WEIGHTS:
            if (op.i[4] == 4) return pc_wino4_weights(P(const float*, 0), op.l[0], op.l[1], op.l[2], op.i[0], op.i[1], op.i[2], op.i[3], P(float*, 1), s);
            return pc_wino_weights(P(const float*, 0), op.l[0], op.l[1], op.l[2], op.i[0], op.i[1], op.i[2], op.i[3], P(float*, 1), s);
        default:
            pc_set_error("pc_run_ops: unknown op kind %d", op.kind);
            return PC_E_ARG;
    }
}

// Per-thread event pool for FORK/JOIN (timing disabled: they only order streams).
static thread_local hipEvent_t g_ev[PC_MAX_LANES];
static thread_local bool g_ev_init = false;

// Deferred kernel timing: event pairs recorded by pc_run_ops_timed(..., ms = NULL) wait here until
// pc_run_ops_timed_collect reads them, so the timed replay adds no host synchronisation of its own.
static thread_local std::vector<hipEvent_t> g_pending;
thread_local hipEvent_t pc_tl_ev_start = nullptr, pc_tl_ev_stop = nullptr;
// timing events are pooled per thread: creating and destroying 212 events per step cost more than recording them
static thread_local std::vector<hipEvent_t> g_ev_pool;
static hipEvent_t take_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

static const bool g_no_ext_events = getenv("PICONS_TIMED_RECORD") && atoi(getenv("PICONS_TIMED_RECORD"));   // 1: hipEventRecord pairs for conv ops too

static int run_list(const pc_op* ops, int n, const pc_stream* lanes, int nlanes, int kind, float* ms, int* count) {
    if (!ops && n > 0) { pc_set_error("pc_run_ops: null ops"); return PC_E_ARG; }
    if (nlanes < 1 || nlanes > PC_MAX_LANES || !lanes) { pc_set_error("pc_run_ops: nlanes=%d (1..%d)", nlanes, PC_MAX_LANES); return PC_E_ARG; }
    if (nlanes > 1 && !g_ev_init) {
        for (int i = 0; i < PC_MAX_LANES; ++i)
            if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) { pc_set_error("hipEventCreate failed"); return PC_E_LAUNCH; }
        g_ev_init = true;
    }
    int cnt = 0;
    hipEvent_t* ev = nullptr;
    // kind = op kind | (sub << 16).  PC_OP_WGRAD: sub 1 = the launches that multiply on the bf16 matrix cores (PC_WG_X6 on the row-segment /
    // generic / stem routes: pc_wgrad_uses_x6), sub 2 = the fp32-MFMA ones (9-tap spectral planes, un-flagged problems), 0 = all
    const int sub = kind > 0 ? kind >> 16 : 0;
    if (kind > 0) kind &= 0xffff;
    auto timed_op = [&](const pc_op& op) {
        if (kind <= 0 || op.kind != kind) return false;
        if (!sub || kind != PC_OP_WGRAD) return true;
        pc_wgrad_desc d;
        memcpy(&d, op.i, sizeof(d));
        const bool x6 = pc_wgrad_uses_x6(&d) != 0;
        return sub == 1 ? x6 : !x6;
    };
    if (kind > 0) {
        for (int k = 0; k < n; ++k) cnt += timed_op(ops[k]);
        ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * (cnt > 0 ? cnt : 1));
        for (int i = 0; i < 2 * cnt; ++i) ev[i] = take_event();
    }
    int j = 0, rc = PC_OK, k = 0;
    for (; k < n && rc == PC_OK; ++k) {
        const pc_op& op = ops[k];
        if (op.kind == PC_OP_FORK || op.kind == PC_OP_JOIN) {
            if (nlanes == 1) continue;
            const bool fork = op.kind == PC_OP_FORK;
            // FORK: the lanes in mask i[0] wait for everything enqueued so far on lane i[1] (0 unless the plan says otherwise)
            const int src = (fork && op.i[1] > 0 && op.i[1] < nlanes) ? op.i[1] : 0;
            // a failed record / wait would silently drop a dependency between lanes (a race, not an error): fail the list instead
            hipError_t he = hipSuccess;
            if (fork) he = hipEventRecord(g_ev[src], (hipStream_t)lanes[src]);
            for (int q = 0; q < nlanes && he == hipSuccess; ++q) {
                if (!((op.i[0] >> q) & 1) || (fork && q == src) || (!fork && q == 0)) continue;
                if (fork) {
                    he = hipStreamWaitEvent((hipStream_t)lanes[q], g_ev[src], 0);
                } else {
                    he = hipEventRecord(g_ev[q], (hipStream_t)lanes[q]);
                    if (he == hipSuccess) he = hipStreamWaitEvent((hipStream_t)lanes[0], g_ev[q], 0);
                }
            }
            if (he != hipSuccess) { pc_set_error("%s: event record / wait failed: %s", fork ? "FORK" : "JOIN", hipGetErrorString(he)); rc = PC_E_LAUNCH; ++k; break; }
            continue;
        }
        const int ln = (op.lane > 0 && op.lane < nlanes) ? op.lane : 0;
        const pc_stream s = lanes[ln];
        const bool t = timed_op(op);
        // a conv op is exactly one kernel: its event pair rides in the dispatch itself (no extra packets on the stream);
        // any other kind is bracketed by recorded events.  (A weight-gradient op is one kernel too, except the direct PrimaryCaps form's
        // two row ranges, where the pair brackets the last of the two launches.)
        const bool ext = t && (op.kind == PC_OP_CONV || op.kind == PC_OP_CONV_X6 || op.kind == PC_OP_WINO_CONV || op.kind == PC_OP_WGRAD) && !g_no_ext_events;
        if (ext) { pc_tl_ev_start = ev[2 * j]; pc_tl_ev_stop = ev[2 * j + 1]; }
        else if (t) (void)hipEventRecord(ev[2 * j], (hipStream_t)s);
        rc = run_one(op, s);
        if (ext) { pc_tl_ev_start = pc_tl_ev_stop = nullptr; ++j; }
        else if (t) { (void)hipEventRecord(ev[2 * j + 1], (hipStream_t)s); ++j; }
    }
    if (rc != PC_OK) {
        char tmp[400];
        snprintf(tmp, sizeof(tmp), "%s", g_err);
        pc_set_error("op %d (kind %d): %s", k - 1, ops[k - 1].kind, tmp);
    }
    if (kind > 0 && !ms) {              // deferred: keep the recorded pairs for pc_run_ops_timed_collect
        for (int i = 0; i < 2 * j; ++i) g_pending.push_back(ev[i]);
        for (int i = 2 * j; i < 2 * cnt; ++i) g_ev_pool.push_back(ev[i]);
        free(ev);
        if (count) *count = j;
    } else if (kind > 0) {
        for (int q = 0; q < nlanes; ++q) (void)hipStreamSynchronize((hipStream_t)lanes[q]);
        float total = 0.f;
        for (int i = 0; i < j; ++i) { float e = 0.f; (void)hipEventElapsedTime(&e, ev[2 * i], ev[2 * i + 1]); total += e; }
        for (int i = 0; i < 2 * cnt; ++i) g_ev_pool.push_back(ev[i]);
        free(ev);
        if (ms) *ms = total;
        if (count) *count = j;
    }
    return rc;
}

extern "C" int pc_run_ops(const pc_op* ops, int n, pc_stream s) { return run_list(ops, n, &s, 1, 0, nullptr, nullptr); }

extern "C" int pc_run_ops_lanes(const pc_op* ops, int n, const pc_stream* lanes, int nlanes) {
    return run_list(ops, n, lanes, nlanes, 0, nullptr, nullptr);
}

static thread_local hipEvent_t g_fan_ev[PC_MAX_LANES];
static thread_local bool g_fan_init = false;

extern "C" int pc_streams_fanin(pc_stream target, const pc_stream* lanes, int nlanes) {
    if (nlanes < 0 || nlanes > PC_MAX_LANES || (nlanes && !lanes)) { pc_set_error("pc_streams_fanin: nlanes=%d (0..%d)", nlanes, PC_MAX_LANES); return PC_E_ARG; }
    hipEvent_t* ev = g_fan_ev;
    if (!g_fan_init) {
        for (int i = 0; i < PC_MAX_LANES; ++i)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { pc_set_error("hipEventCreate failed"); return PC_E_LAUNCH; }
        g_fan_init = true;
    }
    for (int q = 0; q < nlanes; ++q) {
        if (lanes[q] == target) continue;
        if (hipEventRecord(ev[q], (hipStream_t)lanes[q]) != hipSuccess || hipStreamWaitEvent((hipStream_t)target, ev[q], 0) != hipSuccess) {
            pc_set_error("pc_streams_fanin: event record / wait failed on lane %d", q);
            return PC_E_LAUNCH;
        }
    }
    return PC_OK;
}

// Times every op of `kind` with a hipEvent pair on the SAME stream that op's kernels run on
// (torch.cuda.Event would only see torch's current stream).  Synchronises at the end.
extern "C" int pc_run_ops_timed(const pc_op* ops, int n, int kind, float* ms, int* count, const pc_stream* lanes, int nlanes) {
    if (kind <= 0) { pc_set_error("pc_run_ops_timed: kind=%d", kind); return PC_E_ARG; }
    return run_list(ops, n, lanes, nlanes, kind, ms, count);
}

// The calling thread's events (FORK / JOIN, fan-in, the timing pool) are created on first use and kept; this gives them back.
// Pending timed pairs are dropped unread.  The next call that needs events creates new ones.
extern "C" int pc_release_thread_events(void) {
    if (g_ev_init) { for (int i = 0; i < PC_MAX_LANES; ++i) (void)hipEventDestroy(g_ev[i]); g_ev_init = false; }
    if (g_fan_init) { for (int i = 0; i < PC_MAX_LANES; ++i) (void)hipEventDestroy(g_fan_ev[i]); g_fan_init = false; }
    for (hipEvent_t e : g_pending) (void)hipEventDestroy(e);
    for (hipEvent_t e : g_ev_pool) (void)hipEventDestroy(e);
    g_pending.clear(); g_ev_pool.clear();
    return PC_OK;
}

extern "C" int pc_run_ops_timed_collect(float* ms, int* count) {
    float total = 0.f;
    const int pairs = (int)g_pending.size() / 2;
    if (pairs) (void)hipEventSynchronize(g_pending.back());
    static const char* dump = getenv("PICONS_TIMED_DUMP");       // diagnostics: one line per collected pair (ms) appended to this file
    FILE* df = dump ? fopen(dump, "a") : nullptr;
    if (df) fprintf(df, "# collect %d pairs\n", pairs);
    for (int i = 0; i < pairs; ++i) {
        float e = 0.f;
        (void)hipEventSynchronize(g_pending[2 * i + 1]);
        (void)hipEventElapsedTime(&e, g_pending[2 * i], g_pending[2 * i + 1]);
        if (df) fprintf(df, "%d %.4f\n", i, e);
        total += e;
        g_ev_pool.push_back(g_pending[2 * i]);
        g_ev_pool.push_back(g_pending[2 * i + 1]);
    }
    g_pending.clear();
    if (df) fclose(df);
    if (ms) *ms = total;
    if (count) *count = pairs;
    return PC_OK;
}
