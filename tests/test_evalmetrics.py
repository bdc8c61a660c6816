"""f-mAP / v-mAP evaluation (reference loop: evaluate_ucf101.py:73-186).

CPU: the oracle restatement against the accumulators of the reference's own loop (tests/golden/eval_map.npz, made by
tools/make_eval_golden.py from the synthetic videos of tests/evalfixture.py), and the vectorised clip builder against
the oracle's literal one.  GPU: the HIP accumulation (pc_seg_frame_counts / pc_map_accumulate through the C-ABI) against
both, bit-exact -- the tables are integers."""
import os

import numpy as np
import pytest
import torch

from oracle import evalmetrics as oe
from tests import evalfixture as ef

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_map.npz")


@pytest.fixture(scope="module")
def vids():
    return ef.videos()


def test_oracle_matches_reference_loop(vids):
    G = np.load(GOLD)
    st = oe.evaluate(ef.FakeNet().eval(), vids)
    for k in ("frame_ious", "video_ious", "n_tot_frames", "n_vids"):
        assert np.array_equal(getattr(st, k), G[k]), k
    assert st.n_correct == int(G["n_correct"])
    r = st.result()
    assert np.array_equal(r["fmAP"], G["fmAP"], equal_nan=True) and np.array_equal(r["vmAP"], G["vmAP"], equal_nan=True)
    assert np.array_equal(st.iou_threshs, G["iou_threshs"])
    assert 0.2 < G["fmAP"][10] < 0.4 and G["vmAP"][4] > 0.9          # the fixture spreads IoUs over the thresholds


def test_clip_builder_matches_literal_loop(vids):
    from picons_amd import evalmetrics as em
    for video, bbox, lab in vids[:5] + vids[-2:]:
        c, b = em.make_clips(video, bbox)
        ref = oe.make_clips(video, bbox, lab)
        assert c.shape[0] == len(ref)
        for i, (v, bb, _l) in enumerate(ref):
            assert np.array_equal(c[i], v) and np.array_equal(b[i], bb)
    assert em.make_clips(vids[-1][0], vids[-1][1])[0].shape[0] == 0   # the video without boxes yields no clip
    # ragged end: 19 frames -> second window starts at 16, frames past the end are zero
    v = np.arange(19, dtype=np.float32)[:, None, None, None] * np.ones((1, 4, 4, 3), np.float32) + 1
    bb = np.ones((19, 4, 4, 1), np.float32)
    c, b = em.make_clips(v, bb)
    assert c.shape[0] == 4 and [int(x) for x in c[2, :, 0, 0, 0]] == [17, 19, 0, 0, 0, 0, 0, 0] and b[3, 1:].sum() == 0


@pytest.mark.gpu
def test_device_clip_builder_matches_host_builder(vids):
    from picons_amd import evalmetrics as em
    for video, bbox, _lab in vids[:4] + vids[-2:]:
        c, b = em.make_clips(video, bbox)
        d, bd = em.make_clips_device(video, bbox)
        assert d.shape[0] == c.shape[0]
        if c.shape[0]:
            assert np.array_equal(d.cpu().numpy(), np.transpose(c, [0, 4, 1, 2, 3])) and np.array_equal(bd.cpu().numpy(), b[..., 0])


@pytest.mark.gpu
def test_hip_accumulation_matches_reference_loop(vids):
    from picons_amd import evalmetrics as em
    G = np.load(GOLD)
    acc = em.evaluate(ef.FakeNet().eval().cuda(), vids)
    r = acc.result()
    for k in ("frame_ious", "video_ious", "n_tot_frames", "n_vids"):
        assert np.array_equal(r[k], G[k]), k
    assert r["n_correct"] == int(G["n_correct"])
    assert np.array_equal(r["fmAP"], G["fmAP"], equal_nan=True) and np.array_equal(r["vmAP"], G["vmAP"], equal_nan=True)
    assert r["accuracy"] == int(G["n_correct"]) / float(G["n_vids"].sum())
    # clips of consecutive videos packed into full batches: same tables
    rp = em.evaluate(ef.FakeNet().eval().cuda(), vids, pack=True).result()
    for k in ("frame_ious", "video_ious", "n_tot_frames", "n_vids"):
        assert np.array_equal(rp[k], G[k]), k
    assert rp["n_correct"] == int(G["n_correct"])


@pytest.mark.gpu
@pytest.mark.parametrize("frames,hw", [(8, 224), (24, 112), (5, 36), (1, 4)])
def test_frame_counts_vs_numpy(frames, hw):
    """Random logits (some exactly 0, +-inf-ish, tiny) and truth with empty frames: counts against numpy on the host."""
    from picons_amd import ops
    g = torch.Generator().manual_seed(frames * 1000 + hw)
    x = torch.randn(frames, hw, hw, generator=g) * 3
    x[0, 0, :4] = torch.tensor([0.0, -0.0, 1e-3, -1e-3])
    x.view(-1)[5:8] = torch.tensor([80.0, -80.0, 200.0])
    gt = (torch.rand(frames, hw, hw, generator=g) < 0.3).float()
    gt[frames // 2] = 0
    c = ops.seg_frame_counts(x.cuda(), gt.cuda()).cpu().numpy()
    pred = (torch.sigmoid(x).numpy() >= 0.5).astype(np.int64)
    s = pred + gt.numpy()
    ref = np.stack([(s == 2).reshape(frames, -1).sum(1), (s != 0).reshape(frames, -1).sum(1), (gt.numpy() != 0).reshape(frames, -1).sum(1)], 1)
    assert np.array_equal(c, ref)
    assert c[frames // 2, 2] == 0 and c[frames // 2, 0] == 0


@pytest.mark.gpu
def test_map_accumulate_threshold_edges():
    """IoUs that sit exactly on a threshold (1/2, 1/5, 1/20 ...) count for it, as `i_over_u >= iou_threshs[k]` does with a
    float64 ratio against float32 thresholds; frames without truth are skipped; bad labels fail loudly."""
    from picons_amd import ops
    counts = torch.tensor([[1, 2, 5], [1, 5, 3], [1, 20, 1], [0, 7, 2], [9, 9, 9], [3, 0, 0], [19, 20, 4], [7, 20, 4]], dtype=torch.int32)
    z = lambda *s: torch.zeros(*s, dtype=torch.int32, device="cuda")
    fh, vh, nf, nv = z(3, 20), z(3, 20), z(3), z(3)
    ops.map_accumulate(counts.cuda(), 1, fh, vh, nf, nv)
    thr = (np.arange(0, 20, dtype=np.float32) / 20).astype(np.float64)      # float64 comparison: oracle/evalmetrics.py header
    ref = np.zeros(20)
    vi = vu = 0
    for i, u, g_ in counts.numpy():
        if g_ == 0:
            continue
        ref += (int(i) / int(u)) >= thr
        vi += int(i); vu += int(u)
    assert np.array_equal(fh.cpu().numpy()[1], ref) and fh.cpu().numpy()[[0, 2]].sum() == 0
    assert np.array_equal(vh.cpu().numpy()[1], ((vi / vu) >= thr).astype(np.int32))
    assert nf.cpu().tolist() == [0, 7, 0] and nv.cpu().tolist() == [0, 1, 0]
    ops.map_accumulate(counts.cuda(), 1, fh, vh, nf, nv)          # tables accumulate
    assert np.array_equal(fh.cpu().numpy()[1], 2 * ref) and nv.cpu().tolist() == [0, 2, 0]
    with pytest.raises(RuntimeError, match="label"):
        ops.map_accumulate(counts.cuda(), 3, fh, vh, nf, nv)


@pytest.mark.gpu
def test_evaluate_with_the_real_network_teacher_forced():
    """The whole eval path with the HIP CapsNet in eval mode (running-stat BN, argmax class capsule) on two synthetic
    videos: the device tables equal the oracle's accumulation of the SAME logits (recorded from the HIP forward), so clip
    order, the (B,1,8,H,W) frame order against the truth frames and the ragged last batch are all exercised."""
    from picons_amd import evalmetrics as em, model as pmodel, synthetic
    hw = 112
    m = pmodel.CapsNet(pt_path=None, hw=hw, init="conditioned").cuda()
    m.eval(); m.training = False
    vids = synthetic.make_eval_videos(2, seed=3, hw=hw)
    rec = []

    def recording(data, a, b, e, t):
        out = m(data, a, b, e, t)
        rec.append((out[0].detach().cpu().numpy(), out[1].detach().cpu().numpy()))
        return out
    acc = em.evaluate(recording, vids, clip_batch_size=3)
    r = acc.result()
    st = oe.MapState(24)
    it = iter(rec)
    for video, bbox, label in vids:
        clips = oe.make_clips(video, bbox, label)
        segs, preds = [], []
        for i in range(0, len(clips), 3):
            s_, p_ = next(it)
            assert s_.shape == (min(3, len(clips) - i), 1, 8, hw, hw)
            segs.append(s_); preds.append(p_)
        gt = np.stack([c[1] for c in clips]).reshape(-1, hw, hw, 1)
        st.add_video(np.concatenate(segs), gt, np.concatenate(preds), label)
    assert np.array_equal(r["frame_ious"], st.frame_ious) and np.array_equal(r["video_ious"], st.video_ious)
    assert np.array_equal(r["n_tot_frames"], st.n_tot_frames) and np.array_equal(r["n_vids"], st.n_vids) and r["n_correct"] == st.n_correct
    assert r["n_tot_frames"].sum() > 0


@pytest.mark.gpu
def test_dropin_evaluate_cli(tmp_path, monkeypatch, capsys):
    """dropin/evaluate_ucf101.py: the reference's flags and checkpoint policy on synthetic videos."""
    import sys
    from picons_amd import model as pmodel
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi-consistency-activity-detection_amd", "dropin")
    monkeypatch.syspath_prepend(d)
    monkeypatch.setenv("PICONS_SYNTHETIC", "1"); monkeypatch.setenv("PICONS_EVAL_VIDEOS", "1")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k in ("models", "utils") or k.startswith(("models.", "utils."))}
    try:
        m = pmodel.CapsNet(pt_path=None, init="conditioned")
        for tag in ("a", "b"):
            torch.save(m.state_dict(), str(tmp_path / ("best_model_train_%s.pth" % tag)))
        del m
        torch.cuda.empty_cache()
        import evaluate_ucf101
        res = evaluate_ucf101.iou('train', ["--ckpt", str(tmp_path)])
        assert len(res) == 2 and np.array_equal(res[0]["frame_ious"], res[1]["frame_ious"])
        out = capsys.readouterr().out
        assert out.count("Accuracy:") == 2 and "IoU/fmap/vmap" in out
        assert sorted(os.listdir(str(tmp_path))) == ["best_model_train_a.pth"]      # the tie goes to the first; the other is pruned
    finally:
        for k in list(sys.modules):
            if k in ("models", "utils", "evaluate_ucf101") or k.startswith(("models.", "utils.")):
                del sys.modules[k]
        sys.modules.update(saved)
