"""CPU: structure of a multi-lane plan (the Inception branches on separate HIP streams between
FORK/JOIN ops).  The runner trusts the plan, so the plan's invariants are checked here:
balanced regions, lane tags only inside them, no buffer written by one lane and touched by
another inside a region, gradient buckets ready only at joined points, and the same op multiset
as the single-lane plan."""
import collections

import pytest

from picons_amd import capi, step as pstep
from picons_amd.plan import Plan

# p[] slots each op kind writes (the kinds that may appear inside a FORK..JOIN region)
WRITES = {capi.OP_CONV: (4, 5), capi.OP_WGRAD: (2,), capi.OP_BN_FINALIZE: (3, 4, 5), capi.OP_BN_APPLY: (2,),
          capi.OP_BN_BWD: (3, 4, 5, 6), capi.OP_POOL_FWD: (1, 2), capi.OP_POOL_BWD: (2,), capi.OP_TRANSPOSE: (1,),
          capi.OP_FILL: (0,), capi.OP_BN_EVAL_STAT: (4,), capi.OP_WGRAD: (2,), capi.OP_WSPEC_MASTER_FWD: (2, 3)}


def _plan(lanes, bs=1, hw=112):
    args = pstep.default_args(bv=True, n_frames=5)
    p = Plan(24, hw, n=bs, groups=2, lanes=lanes)
    p.build_forward(); p.build_loss(args); p.build_backward(); p.build_adam()
    p.finalize()
    return p


def _regions(lst):
    """-> [(fork_idx, join_idx)], asserting balance and that lane tags only occur inside regions."""
    out, open_at = [], None
    for idx, op in enumerate(lst):
        kind, lane = op[0], op[5]
        if kind == capi.OP_FORK:
            assert open_at is None, "nested FORK at %d" % idx
            open_at = idx
        elif kind == capi.OP_JOIN:
            assert open_at is not None, "JOIN without FORK at %d" % idx
            out.append((open_at, idx)); open_at = None
        else:
            assert lane == 0 or open_at is not None, "op %d on lane %d outside a FORK..JOIN region" % (idx, lane)
    assert open_at is None, "list ends with an open FORK"
    return out


@pytest.mark.parametrize("lanes", [2, 4])
def test_regions_balanced_and_private(lanes):
    p = _plan(lanes)
    for name in ("prep", "fwd", "loss", "bwd", "adam"):
        lst = p.lists[name]
        regs = _regions(lst)
        if name in ("fwd", "bwd"):
            assert len(regs) == 8, "one region per Inception module (Mixed_3b..4f) + the merged tail's position classes"
        elif name == "prep":
            assert len(regs) == 1, "weight-layout prep: one region"
        else:
            assert not regs
        for a, b in regs:
            touched = collections.defaultdict(set)   # ref -> lanes that touch it
            written = collections.defaultdict(set)   # ref -> lanes that write it
            for op in lst[a + 1:b]:
                kind, ptrs, lane = op[0], op[3], op[5]
                assert kind in WRITES, "op kind %d inside a lane region has no write map" % kind
                for q, r in enumerate(ptrs):
                    if r is None:
                        continue
                    if kind == capi.OP_CONV and q == 4:
                        # a conv writes the sub-lattice (ooff, extents) of its output tensor: the merged tail's position classes
                        # tile one buffer with disjoint sub-lattices on two lanes
                        i = op[1]
                        r = (r, tuple(i[6:9]), tuple(i[17:20]))
                    touched[r].add(lane)
                    if q in WRITES[kind]:
                        written[r].add(lane)
            for r, wl in written.items():
                assert len(touched[r]) == 1, "buffer %s written on lane(s) %s but touched on %s" % (r, wl, touched[r])


def test_same_ops_as_single_lane_plan():
    p1, p4 = _plan(1), _plan(4)
    assert p1.arena_bytes == p4.arena_bytes
    for name in ("prep", "fwd", "loss", "bwd", "adam"):
        strip = lambda lst: collections.Counter((op[0], tuple(op[1]), tuple(op[2]), tuple(op[4])) for op in lst
                                                if op[0] not in (capi.OP_FORK, capi.OP_JOIN))
        assert strip(p1.lists[name]) == strip(p4.lists[name]), name
    assert all(op[5] == 0 for lst in p1.lists.values() for op in lst)


def test_buckets_ready_only_at_joined_points():
    p = _plan(4)
    regs = _regions(p.lists["bwd"])
    b = p.grad_buckets(500_000)
    assert len(b) > 8
    for ready, _a, _e in b:
        assert not any(f < ready <= j for f, j in regs), "bucket ready inside an open region (%d)" % ready
    spans = sorted((a, e) for _r, a, e in b)
    assert spans[0][0] == 0 and spans[-1][1] == p.nparams
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))


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
    assert count(q, capi.OP_CONV) - count(p, capi.OP_CONV) == 14 and count(q, capi.OP_BN_APPLY) - count(p, capi.OP_BN_APPLY) == 14
