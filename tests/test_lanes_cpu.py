"""CPU: structure of a multi-lane plan (the Inception branches on separate HIP streams between
FORK/JOIN ops).  The runner trusts the plan, so the plan's invariants are checked here:
balanced regions, lane tags only inside them, no buffer written by one lane and touched by
another inside a region, gradient buckets ready only at joined points, and the same op multiset
as the single-lane plan."""
import collections

import pytest

from picons_amd import capi, step as pstep
from picons_amd.plan import Plan

# p[] slots each op kind WRITES (every other non-null slot is read); kinds are those plan.py emits into fwd / loss / bwd
WRITES = {capi.OP_CONV: (4, 5), capi.OP_WGRAD: (2,), capi.OP_BN_FINALIZE: (3, 4, 5, 6), capi.OP_BN_APPLY: (2,), capi.OP_BN_EVAL_STAT: (4,),
          capi.OP_BN_FIN_APPLY: (3, 4, 5, 7),
          capi.OP_BN_BWD: (3, 4, 5, 6), capi.OP_POOL_FWD: (1, 2), capi.OP_POOL_BWD: (2,), capi.OP_CHSCALE: (2,), capi.OP_ACT_BWD: (2, 3, 4),
          capi.OP_TO_NDHWC: (1,), capi.OP_TRANSPOSE: (1,), capi.OP_FILL: (0,), capi.OP_EM_FWD: (4, 5), capi.OP_EM_BWD: (5, 6, 7, 8, 9),
          capi.OP_CMASK_FWD: (3, 4, 5), capi.OP_CMASK_BWD: (3,), capi.OP_TAIL_COMBINE: (4, 5, 6), capi.OP_TAIL6_WEIGHTS: (1, 2),
          capi.OP_TAIL6_GATHER: (3,), capi.OP_TAIL6_SCATTER: (1,), capi.OP_TAIL6_BIAS_SUMS: (1, 2), capi.OP_TAIL6_WGRAD_MAP: (1,),
          capi.OP_TAIL_GRADS: (6, 7, 8, 9, 10), capi.OP_AXIS: (3,), capi.OP_WSPEC_FWD: (2,), capi.OP_WSPEC_BWD: (2,), capi.OP_WSPEC_MASTER_FWD: (2, 3),
          capi.OP_WSPEC_MASTER_BWD: (2,), capi.OP_LOSS: (4, 5, 6, 7, 8, 9), capi.OP_SPREAD: (3, 4), capi.OP_ADAM: (0, 2, 3), capi.OP_COL2IM: (1,),
          capi.OP_TAPSUM_FWD: (2,), capi.OP_TAPSUM_BWD: (1,), capi.OP_TAIL_COLSUM: (1,), capi.OP_WINO_CONV: (3, 4), capi.OP_WINO_WEIGHTS: (1,),
          capi.OP_CONV_X6: (4, 5, 6), capi.OP_SPLIT_PLANES: (1,), capi.OP_WSPEC_MASTER_PLANES: (2, 3), capi.OP_WGRAD_FOLD: (0,)}


def _plan(lanes, bs=1, hw=112, early_adam=False, exp=None):
    args = pstep.default_args(bv=True, n_frames=5)
    p = Plan(24, hw, n=bs, groups=2, lanes=lanes, early_adam=early_adam, exp=exp)
    p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam()
    p.finalize()
    return p


def _regions(lst):
    """-> [(fork_idx, join_idx)] of a list whose regions do not nest (the prep list)."""
    out, open_at = [], None
    for idx, op in enumerate(lst):
        if op[0] == capi.OP_FORK:
            assert open_at is None, "nested FORK at %d" % idx
            open_at = idx
        elif op[0] == capi.OP_JOIN:
            assert open_at is not None, "JOIN without FORK at %d" % idx
            out.append((open_at, idx)); open_at = None
        else:
            assert op[5] == 0 or open_at is not None, "op %d on lane %d outside a FORK..JOIN region" % (idx, op[5])
    assert open_at is None, "list ends with an open FORK"
    return out


def _accesses(p, op):
    """-> [(buffer key, is_write)] of one op.  Keys are the start references the plan handed out (channel slices of one buffer
    have different keys and never overlap); a conv's output key carries its sub-lattice (the merged tail's position classes tile
    one buffer with disjoint sub-lattices)."""
    kind, ints, ptrs = op[0], op[1], op[3]
    out = []
    if kind == capi.OP_TRANSPOSE_MULTI:
        for job in p.multi_jobs[ptrs[0][1]]:
            out += [(job[3][0], False), (job[3][1], True)]
        return out
    if kind == capi.OP_WGRAD_MULTI:
        for _d, refs in p.wjobs[ptrs[0][1]]:
            out += [(refs[0], False), (refs[1], False), (refs[2], True)]
        return out
    assert kind in WRITES, "op kind %d has no write map" % kind

    def weight_base(r):
        """A reference into a registered kernel-layout weight buffer (a group's / a position class's slice) or into its bf16 planes ->
        the buffer's start reference, which is the key its producer (transpose, split) writes under."""
        for b in p.wbufs:
            if b["ref"][0] == r[0] and b["ref"][1] <= r[1] < b["ref"][1] + 4 * b["n"]:
                return b["ref"]
            pl = b["planes"]
            if pl is not None and pl[0] == r[0] and pl[1] <= r[1] < pl[1] + 6 * b["n"]:
                return pl
        return r
    for q, r in enumerate(ptrs):
        if r is None:
            continue
        if kind in (capi.OP_CONV, capi.OP_CONV_X6) and q == 4:
            key = (r, tuple(ints[6:9]), tuple(ints[17:20]))
        elif kind in (capi.OP_CONV, capi.OP_CONV_X6) and q == 1:
            key = weight_base(r)
        else:
            key = r
        out.append((key, q in WRITES[kind]))
    return out


def _check_list(p, lst, lanes):
    """Happens-before over one op list with vector clocks: ops of a lane are ordered; FORK(mask) orders everything lane 0 has
    enqueued before the lanes in mask; JOIN(mask) orders everything those lanes have enqueued before lane 0's next op.  Two ops
    that touch the same buffer, at least one writing it, must be ordered; every side lane must be joined before the list ends."""
    clock = [[0] * lanes for _ in range(lanes)]
    unjoined = set()
    last = {}                                  # key -> [(stamp, lane, idx, is_write)]
    nregions = 0
    for idx, op in enumerate(lst):
        kind, lane = op[0], op[5]
        if kind == capi.OP_FORK:
            src = op[1][1] if len(op[1]) > 1 else 0
            nregions += src == 0
            for q in range(lanes):
                if (op[1][0] >> q) & 1 and q != src:
                    clock[q] = [max(a, b) for a, b in zip(clock[q], clock[src])]
            continue
        if kind == capi.OP_JOIN:
            for q in range(1, lanes):
                if (op[1][0] >> q) & 1:
                    clock[0] = [max(a, b) for a, b in zip(clock[0], clock[q])]
                    unjoined.discard(q)
            continue
        assert 0 <= lane < lanes
        clock[lane][lane] += 1
        stamp = list(clock[lane])
        if lane:
            unjoined.add(lane)
        for key, wr in _accesses(p, op):
            for st2, lane2, idx2, wr2 in last.get(key, ()):
                if (wr or wr2) and lane2 != lane:
                    assert st2[lane2] <= stamp[lane2], "ops %d (lane %d) and %d (lane %d) touch %s unordered" % (idx2, lane2, idx, lane, key)
            last.setdefault(key, []).append((stamp, lane, idx, wr))
    assert not unjoined, "lanes %s still have unjoined work at the end of the list" % sorted(unjoined)
    return nregions


def _forks(p, name, mask):
    return sum(1 for op in p.lists[name] if op[0] == capi.OP_FORK and op[1][0] == mask)


@pytest.mark.parametrize("lanes", [2, 3, 4, 5])
def test_lanes_are_ordered_wherever_they_share_a_buffer(lanes):
    p = _plan(lanes)
    assert bool(p.skip_lane) == (lanes >= 3) and bool(p.wg_lane) == (lanes >= 3)
    assert (p.wg_lane != p.skip_lane) == (lanes >= 4) and p.branch_lanes == {2: 2, 3: 2, 4: 2, 5: 3}[lanes]
    assert p.late_prep == (lanes >= 3)
    # prep_late is enqueued on the side lanes in front of the forward list, which joins it before Mixed_3b: one sequence
    _check_list(p, p.lists["prep_late"] + p.lists["fwd"], lanes)
    for name in ("loss", "bwd", "adam"):
        _check_list(p, p.lists[name], lanes)
    if p.late_prep:
        assert len(p.lists["prep"]) < 12 < len(p.lists["prep_late"]) and all(op[5] != 0 for op in p.lists["prep_late"][1:])
    branch = (1 << p.branch_lanes) - 2
    # backward: one FORK of the branch lanes per Inception module (Mixed_3b..4f) + the merged tail's position classes.  Forward (round 6): a
    # module forks TWICE -- lane 1 at its head (the pool branch needs nothing but the module's input), then, behind the fused 1x1x1 unit, the
    # weight-gradient lane (idle in the forward: the small 3x3x3 branch) or, without one, lane 1 again
    # ... and the forward deals the position classes of upsample2 / upsample3 to ALL lanes (one FORK of every side lane each)
    all_side = (1 << lanes) - 2
    second = (1 << p.wg_lane) if p.wg_lane else branch
    masks = {2, second, branch}
    others = (2 if all_side in masks else 0) + (3 if p.skip_lane and (1 << p.skip_lane) in masks else 0)
    assert _forks(p, "bwd", branch) == 8
    assert sum(_forks(p, "fwd", m) for m in masks) == 7 + 7 + 1 + others
    assert _forks(p, "fwd", all_side) >= 2
    if p.skip_lane and p.skip_lane != p.wg_lane:
        assert _forks(p, "fwd", 1 << p.skip_lane) == 3                   # conv56, conv112, conv28
        assert _forks(p, "bwd", 1 << p.skip_lane) == 2                   # conv56, conv112 (conv28's backward stays on lane 0)
        on_skip = {name: [op for op in p.lists[name] if op[5] == p.skip_lane] for name in ("fwd", "bwd")}
        assert sum(1 for op in on_skip["bwd"] if op[0] in (capi.OP_CONV, capi.OP_WINO_CONV)) == 2
        assert sum(1 for op in on_skip["fwd"] if op[0] == capi.OP_WINO_CONV) == 2
        nclass = sum(1 for op in on_skip["fwd"] if op[0] == capi.OP_CONV) - 1       # conv28 + this lane's share of the 16 position classes
        assert 2 <= nclass <= 6
    if p.wg_lane:
        wg = [op for op in p.lists["bwd"] if op[0] in (capi.OP_WGRAD, capi.OP_WGRAD_MULTI)]
        assert all(op[5] == p.wg_lane for op in wg)
        assert not any(op[0] == capi.OP_WGRAD_MULTI for op in wg)          # grouped launches are opt-in (PICONS_WGRAD_MULTI*)
        assert sum(1 for op in p.lists["fwd"] if op[5] == p.wg_lane and op[0] in (capi.OP_CONV, capi.OP_WINO_CONV)) >= 7


def test_inception_wgrads_as_grouped_launches(monkeypatch):
    monkeypatch.setenv("PICONS_WGRAD_ATOMIC", "1")           # grouped launches exist in the atomic split-K form only (no K-slice workspaces)
    p = _plan(4, exp={"PICONS_WGRAD_MULTI": "1", "PICONS_WGRAD_MULTI_TAIL": "1"})      # experiment switches: passed explicitly (switches.py)
    _check_list(p, p.lists["bwd"], 4)
    assert sorted(len(j) for j in p.wjobs) == [4] * 7 + [8]
    assert sum(1 for op in p.lists["bwd"] if op[0] == capi.OP_WGRAD_MULTI) == 8


def test_side_lanes_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("PICONS_SKIP_LANE", "0"); monkeypatch.setenv("PICONS_WGRAD_LANE", "0")
    p = _plan(3)
    assert not p.skip_lane and not p.wg_lane and p.branch_lanes == 3
    _check_list(p, p.lists["bwd"], 3)
    assert _forks(p, "bwd", 6) == 8
    monkeypatch.setenv("PICONS_SKIP_LANE", "1"); monkeypatch.setenv("PICONS_WGRAD_LANE", "1"); monkeypatch.setenv("PICONS_WGRAD_SEPARATE", "0")
    q = _plan(4)
    assert q.skip_lane == q.wg_lane == 3 and q.branch_lanes == 3
    _check_list(q, q.lists["prep_late"] + q.lists["fwd"], 4)
    _check_list(q, q.lists["bwd"], 4)


def test_same_ops_as_single_lane_plan():
    """Lanes only re-order: the multiset of ops (the gradient re-layout jobs counted one by one, since the lanes group them into
    different multi-job launches) is that of the single-lane plan."""
    p1, p4 = _plan(1), _plan(4)

    def strip(p, lst):
        out = collections.Counter()
        for op in lst:
            if op[0] in (capi.OP_FORK, capi.OP_JOIN):
                continue
            if op[0] == capi.OP_TRANSPOSE_MULTI:
                for job in p.multi_jobs[op[3][0][1]]:
                    out[(capi.OP_TRANSPOSE, tuple(job[1]), tuple(job[2]), tuple(job[4]))] += 1
            elif op[0] == capi.OP_WGRAD_MULTI:
                from picons_amd import desc as D
                for wd, _refs in p.wjobs[op[3][0][1]]:
                    out[(capi.OP_WGRAD, tuple(D.flatten(wd, D.WGRAD_FIELDS)), (), ())] += 1
            else:
                out[(op[0], tuple(op[1]), tuple(op[2]), tuple(op[4]))] += 1
        return out
    assert strip(p1, p1.lists["prep"] + p1.lists["prep_late"]) == strip(p4, p4.lists["prep"] + p4.lists["prep_late"])
    for name in ("fwd", "loss", "bwd", "adam"):
        assert strip(p1, p1.lists[name]) == strip(p4, p4.lists[name]), name
    assert all(op[5] == 0 for lst in p1.lists.values() for op in lst)


@pytest.mark.parametrize("lanes", [2, 3, 4, 5])
def test_buckets_ready_only_at_joined_points(lanes):
    """A bucket may be all-reduced once every gradient in it is final AND ordered before lane 0's position `ready` in the
    backward list: replay the list up to `ready` with vector clocks and check the finalising op of every parameter of the
    bucket happens-before lane 0's clock there."""
    p = _plan(lanes)
    bwd = p.lists["bwd"]
    b = p.grad_buckets(500_000, joined=True)
    assert len(b) > 8
    spans = sorted((a, e) for _r, a, e in b)
    assert spans[0][0] == 0 and spans[-1][1] == p.nparams
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    # clocks after every prefix of the list, and the stamp of every op
    clock = [[0] * lanes for _ in range(lanes)]
    stamp_of, lane0_after = {}, []
    for idx, op in enumerate(bwd):
        kind, lane = op[0], op[5]
        if kind == capi.OP_FORK:
            src = op[1][1] if len(op[1]) > 1 else 0
            for q in range(lanes):
                if (op[1][0] >> q) & 1 and q != src:
                    clock[q] = [max(x, y) for x, y in zip(clock[q], clock[src])]
        elif kind == capi.OP_JOIN:
            for q in range(1, lanes):
                if (op[1][0] >> q) & 1:
                    clock[0] = [max(x, y) for x, y in zip(clock[0], clock[q])]
        else:
            clock[lane][lane] += 1
            stamp_of[idx] = (lane, clock[lane][lane])
        lane0_after.append(list(clock[0]))
    for ready, a, e in b:
        assert 0 < ready <= len(bwd)
        seen = lane0_after[ready - 1]
        for nm, off in p.poff.items():
            if a <= off < e:
                k = p.final_at[nm]                       # ops [0, k) finalise it; the last of them on its lane is the one that counts
                last = max((i for i in range(k) if i in stamp_of and stamp_of[i][0] == p.final_lane.get(nm, 0)), default=None)
                assert last is not None and last < ready, nm
                lane, t = stamp_of[last]
                assert t <= seen[lane], "bucket [%d, %d) ready at %d but %s (lane %d) is not ordered before it" % (a, e, ready, nm, lane)


def test_prep_list_is_spread_over_lanes_at_resolve():
    """resolve() wraps the weight-layout prep in one FORK..JOIN: ops reading master parameters round-robin over
    the lanes (all ops of one weight on one lane), second-level layouts after the join on lane 0."""
    p = _plan(2)
    arrs = p.resolve({"A": 1 << 20, "P": 1 << 30, "G": 1 << 31, "M": 1 << 32, "V": 1 << 33, "R": 1 << 34})
    lst = p.lists["prep"]
    regs = _regions(lst)
    assert len(regs) == 1 and regs[0][0] == 0
    f, j = regs[0]
    lanes_used = {op[5] for op in lst[f + 1:j]}
    assert lanes_used == {0, 1}
    dest_lane = {}
    for op in lst[f + 1:j]:
        kind, ptrs, lane = op[0], op[3], op[5]
        dst = ptrs[0] if kind == capi.OP_FILL else ptrs[1]
        assert kind == capi.OP_FILL or ptrs[0][0] == "P"            # sources inside the region: master parameters only
        assert dest_lane.setdefault(dst, lane) == lane               # FILL + transposes of one buffer: same lane
    for op in lst[j + 1:]:
        assert op[5] == 0 and op[3][0][0] == "A"                     # second-level layouts read prepared buffers
    # in the resolved array every lane's parameter re-layouts are one multi-job launch inside the region
    kinds = [int(k) for k in arrs["prep"]["kind"]]
    assert kinds[0] == capi.OP_FORK and kinds.count(capi.OP_TRANSPOSE_MULTI) == 2 and capi.OP_TRANSPOSE not in kinds[:kinds.index(capi.OP_JOIN)]
    multi = arrs["prep"][[k == capi.OP_TRANSPOSE_MULTI for k in kinds]]
    assert sorted(int(x) for x in multi["lane"]) == [0, 1]
    n_par = sum(1 for op in lst[f + 1:j] if op[0] == capi.OP_TRANSPOSE)
    by_ptr = {t.ctypes.data: t for t in arrs["_tjobs"]}
    assert int(multi["i"][:, 0].sum()) == n_par and sum(len(by_ptr[int(multi["p"][q, 0])]) for q in range(2)) == n_par
    # the backward's per-module gradient re-layouts are multi-job launches too, every table is owned by the resolved dict
    bk = arrs["bwd"][arrs["bwd"]["kind"] == capi.OP_TRANSPOSE_MULTI]
    assert len(bk) >= 7 and all(int(x) in by_ptr and len(by_ptr[int(x)]) == int(n) for x, n in zip(bk["p"][:, 0], bk["i"][:, 0]))
    # idempotent
    p.resolve({"A": 1 << 20, "P": 1 << 30, "G": 1 << 31, "M": 1 << 32, "V": 1 << 33, "R": 1 << 34})
    assert len(p.lists["prep"]) == len(lst)


def test_flat_layout_groups_the_stacked_units(monkeypatch):
    """Stacked 1x1x1 Unit3Ds (plan.inception) need their BN parameters and running statistics adjacent in the flat
    buffers: same names and shapes as the reference order, every float covered exactly once, groups adjacent."""
    p = _plan(1)
    monkeypatch.setenv("PICONS_FUSE1X1", "0")
    q = _plan(1)
    assert list(p.pshape) == list(q.pshape) and p.nparams == q.nparams and p.nrunning == q.nrunning
    assert list(q.poff) == list(q.pshape) and sorted(p.poff) == sorted(q.poff)
    for tab, size in ((p.poff, lambda k: -(-int(__import__("numpy").prod(p.pshape[k])) // 4) * 4),
                      (p.roff, lambda k: p.pshape[k.rsplit(".bn.", 1)[0] + ".bn.weight"][0])):
        o = 0
        for k in sorted(tab, key=tab.get):
            assert tab[k] == o, k
            o += size(k)
    assert len(p.fused_groups) == 7
    for grp in p.fused_groups:
        for tab, sfxs in ((p.poff, (".bn.weight", ".bn.bias")), (p.roff, (".bn.running_mean", ".bn.running_var"))):
            for sfx in sfxs:
                o = tab[grp[0] + sfx]
                for pre in grp:
                    assert tab[pre + sfx] == o
                    o += p.pshape[pre + ".bn.weight"][0]
    # one conv / BN per group instead of three: 2 x 7 fewer convs forward
    count = lambda pl, kind: sum(1 for op in pl.lists["fwd"] if op[0] == kind)
    assert count(q, capi.OP_CONV) - count(p, capi.OP_CONV) == 14 and (count(q, capi.OP_BN_APPLY) + count(q, capi.OP_BN_FIN_APPLY)) - (count(p, capi.OP_BN_APPLY) + count(p, capi.OP_BN_FIN_APPLY)) == 14


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_buckets_ready_once_their_last_writer_is_enqueued(lanes):
    """The default schedule (the collective is ordered behind every lane's stream by GradReducer.launch): a bucket is ready as
    soon as the last op that writes one of its gradients is in the enqueued prefix -- on whatever lane -- and not earlier.  With a
    weight-gradient lane that is joined only at the end, this is the difference between buckets leaving during the backward and
    all of them leaving at its end."""
    p = _plan(lanes)
    bwd = p.lists["bwd"]
    b = p.grad_buckets(500_000)
    spans = sorted((a, e) for _r, a, e in b)
    assert spans[0][0] == 0 and spans[-1][1] == p.nparams and all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    gkeys = {p.G(nm): nm for nm in p.pshape}
    last_writer = {}
    for idx, op in enumerate(bwd):
        if op[0] in (capi.OP_FORK, capi.OP_JOIN):
            continue
        for key, wr in _accesses(p, op):
            if wr and key in gkeys:
                last_writer[gkeys[key]] = idx
    for ready, a, e in b:
        inside = [nm for nm, off in p.poff.items() if a <= off < e]
        need = max(last_writer[nm] for nm in inside if nm in last_writer) + 1
        assert need <= ready <= len(bwd), (a, e, ready, need)
        assert ready == max(p.final_at[nm] for nm in inside)
    readies = sorted(r for r, _a, _e in b)
    assert readies[0] < 0.35 * len(bwd) and readies[len(readies) // 2] < 0.9 * len(bwd)       # buckets leave throughout the backward
    if lanes >= 4:       # what the joined schedule would do with a weight-gradient lane: everything at the end of the list
        assert all(r == len(bwd) for r, _a, _e in p.grad_buckets(500_000, joined=True))


def test_frame_size_is_checked_where_the_reference_would_fail():
    """hw / 8 is the feature map the 9x9 valid PrimaryCaps conv runs on (capsules_ucf101.py:43-49): below 72 (or not a multiple of 8)
    the reference dies inside nn.Conv2d; the plan says so instead of failing in a kernel launch."""
    for hw in (64, 100, 60):
        with pytest.raises(ValueError, match="frame size"):
            Plan(24, hw, n=1, groups=2)
    Plan(24, 72, n=1, groups=2)


@pytest.mark.parametrize("lanes", [2, 4])
def test_early_adam_op_is_ordered_behind_every_other_gradient(lanes):
    """StepEngine's plan: one Adam op inside the backward list, over every parameter but the stem's, on a side lane in front of the stem's
    backward.  Every op that finalises a gradient in its range, on whatever lane, must happen-before it; nothing behind it may read a
    parameter in its range; the lane is joined before the list ends."""
    p = _plan(lanes, early_adam=True)
    bwd = p.lists["bwd"]
    k = p.op_adam_early
    assert k is not None and bwd[k][0] == capi.OP_ADAM and bwd[k][4] == [0] and bwd[k][5] != 0          # un-armed, on a side lane
    _check_list(p, bwd, lanes)
    n0 = p.adam_split
    assert n0 == p.poff["conv1.Conv3d_2b_1x1.conv3d.weight"]
    # happens-before of every earlier op: replay the clocks up to k
    clock = [[0] * lanes for _ in range(lanes)]
    stamp_of = {}
    for idx, op in enumerate(bwd[:k + 1]):
        kind, lane = op[0], op[5]
        if kind == capi.OP_FORK:
            src = op[1][1] if len(op[1]) > 1 else 0
            for q in range(lanes):
                if (op[1][0] >> q) & 1 and q != src:
                    clock[q] = [max(a, b) for a, b in zip(clock[q], clock[src])]
        elif kind == capi.OP_JOIN:
            for q in range(1, lanes):
                if (op[1][0] >> q) & 1:
                    clock[0] = [max(a, b) for a, b in zip(clock[0], clock[q])]
        else:
            clock[lane][lane] += 1
            stamp_of[idx] = (lane, list(clock[lane]))
    alane, astamp = stamp_of[k]
    for idx, (lane, st) in stamp_of.items():
        if idx != k:
            assert st[lane] <= astamp[lane], "op %d on lane %d is not ordered before the early Adam" % (idx, lane)
    # every gradient outside the stem is final before it; the ops behind it touch no parameter / gradient / moment in its range
    for nm in p.pshape:
        if p.poff[nm] >= n0:
            assert p.final_at[nm] <= k, nm
    for op in bwd[k + 1:]:
        for r in op[3]:
            if isinstance(r, tuple) and r[0] in ("P", "G", "M", "V"):
                assert r[1] < 4 * n0, (op[0], r)


@pytest.mark.parametrize("classes,bs,hw", [(24, 8, 224), (21, 8, 224), (24, 2, 112)])
def test_no_weight_gradient_of_the_product_plan_is_summed_with_atomics(classes, bs, hw):
    """Round 6: every gradient of the step is bit-identical from run to run because no launch adds fp32 partials in arrival order.  The GPU
    determinism test runs at bs = 2, 112^2, where the K-slice counts differ from the benchmark's; this walks the plans themselves (host only):
    every weight-gradient launch either makes ONE K slice per problem (each element written or added once), or stores plainly (splitk = -1), or
    writes K-slice images (ws_slices >= the slices the launch makes); the tail's three reductions carry their workspaces; no grouped launch
    (its epilogue is atomic) is left."""
    from picons_amd import desc as D
    from picons_amd.plan import _wdesc
    args = pstep.default_args(bv=True, gv=True, n_frames=5)
    p = Plan(classes, hw, n=bs, groups=2, lanes=4, early_adam=True, jhmdb=classes == 21)
    p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam()
    p.finalize()
    vec3 = set(D.CONV_VEC3) | {"doff"}
    seen = collections.Counter()
    for op in p.lists["bwd"]:
        kind, ints, ptrs = op[0], op[1], op[3]
        assert kind != capi.OP_WGRAD_MULTI, "grouped weight-gradient launches add their K slices with atomics"
        if kind == capi.OP_WGRAD:
            d, q = {}, 0
            for f in D.WGRAD_FIELDS:
                n = 3 if f in vec3 else 1
                d[f] = [int(v) for v in ints[q:q + 3]] if n == 3 else int(ints[q])
                q += n
            ns = capi.lib().pc_wgrad_slices(_wdesc(dict(d, ws_slices=0)))
            assert ns >= 1, capi.lib().pc_last_error()
            how = "stores" if d["splitk"] == -1 else ("one slice" if ns == 1 else "images")
            assert how != "images" or d["ws_slices"] >= ns, ("split-K launch without K-slice images", d, ns)
            seen[how] += 1
        elif kind == capi.OP_TAIL6_WGRAD_MAP:
            assert ints[2] == 1 and sum(ints[3:11]) >= 8, ints[:11]
            seen["tail map"] += 1
        elif kind == capi.OP_TAIL6_BIAS_SUMS:
            assert ptrs[2], "pc_tail6_bias_sums without its partial rows"
            seen["tail bias"] += 1
        elif kind == capi.OP_TAIL_GRADS:
            assert ptrs[10], "pc_tail_grads without its block partials"
            seen["tail grads"] += 1
    assert seen["images"] >= 30 and seen["tail map"] == seen["tail bias"] == seen["tail grads"] == 1, seen
