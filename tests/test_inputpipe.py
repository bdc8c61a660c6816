"""Input pipeline (reference: datasets/ucf_dataloader.py:84-264).

CPU: the oracle restatement against samples of the reference's own `__getitem__` (tests/golden/input_pipe.npz, made by
tools/make_input_golden.py on the synthetic decoded videos of tests/inputfixture.py) and the host decisions of the
product module against the oracle.  GPU: `pc_clip_from_u8` through the C-ABI against both -- the float32 outputs are the
reference's float64 values rounded once, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import inputpipe as oi
from tests import inputfixture as fx

GOLD = os.path.join(os.path.dirname(__file__), "golden", "input_pipe.npz")


def _sums(d, a, m, s):
    return np.array([d.sum(), a.sum(), m.sum(), float(s["action"][0]), float(s["label_vid"])])


@pytest.mark.parametrize("k", range(fx.N_CASES))
def test_oracle_matches_reference_loader(k):
    G = np.load(GOLD)
    frames, ann, train = fx.case(k)
    np.random.seed(1000 + k)
    s = oi.get_item(frames, ann, train)
    d, a, m = s["data"].numpy(), s["aug_data"].numpy(), s["loc_msk"].numpy()
    assert d.shape == (3, 8, 224, 224) and m.shape == (1, 8, 224, 224) and d.dtype == np.float64
    assert np.array_equal(d[:, :, ::9, ::7], G["data_%d" % k]) and np.array_equal(a[:, :, ::9, ::7], G["aug_%d" % k])
    assert np.array_equal(np.packbits(m.astype(np.uint8)), G["mask_%d" % k])
    assert np.array_equal(_sums(d, a, m, s), G["sums_%d" % k])


def test_host_decisions_follow_the_reference_draw_order():
    """Same seed -> same annotated frame, window and crop as the oracle (whose draws are the reference's)."""
    from picons_amd import inputpipe as ip
    for k in (0, 1, 2, 3, 4, 5, 9):
        frames, ann, train = fx.case(k)
        np.random.seed(1000 + k)
        per_frame, label, annot_frames, lv = ip.frame_boxes(ann, frames.shape[0])
        span = ip.choose_window(annot_frames, frames.shape[0])
        np.random.seed(1000 + k)
        bbox, label2, annot2, lv2 = oi.rasterise(ann, *frames.shape[:3])
        assert (label, annot_frames, lv) == (label2, annot2, lv2) and span is not None and len(span) == 8
        for f in range(frames.shape[0]):          # the boxes listed for a frame paint exactly the oracle's mask
            m = np.zeros(frames.shape[1:3], np.uint8)
            for x, y, bw, bh in per_frame.get(f, []):
                m[y:y + bh, x:x + bw] = 1
            assert np.array_equal(m, bbox[f, :, :, 0])
    assert ip.choose_window([], 20) is None and ip.choose_window([24], 20) is None
    assert list(ip.choose_window([5], 30)) == [1, 2, 3, 4, 5, 6, 7, 8] and list(ip.choose_window([1], 30)) == list(range(8))
    assert list(ip.choose_window([22], 25)) == [9, 11, 13, 15, 17, 19, 21, 23]


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(fx.N_CASES))
def test_device_pipeline_matches_reference_loader(k):
    from picons_amd import inputpipe as ip
    G = np.load(GOLD)
    frames, ann, train = fx.case(k)
    np.random.seed(1000 + k)
    s = ip.get_item(frames, ann, train)
    np.random.seed(1000 + k)
    o = oi.get_item(frames, ann, train)
    d, a, m = s["data"].cpu().numpy(), s["aug_data"].cpu().numpy(), s["loc_msk"].cpu().numpy()
    assert d.dtype == np.float32 and d.shape == (3, 8, 224, 224) and m.shape == (1, 8, 224, 224)
    assert np.array_equal(d, o["data"].numpy().astype(np.float32)) and np.array_equal(a, o["aug_data"].numpy().astype(np.float32))
    assert np.array_equal(m, o["loc_msk"].numpy().astype(np.float32))
    assert np.array_equal(d[:, :, ::9, ::7], G["data_%d" % k].astype(np.float32)) and np.array_equal(np.packbits(m.astype(np.uint8)), G["mask_%d" % k])
    assert float(s["action"][0]) == G["sums_%d" % k][3] and float(s["label_vid"]) == G["sums_%d" % k][4]


@pytest.mark.gpu
def test_device_frames_and_bad_arguments():
    """Frames already in HBM (whole video, real frame ids), boxes that numpy would clip at the border, and loud failures."""
    from picons_amd import ops
    g = torch.Generator().manual_seed(3)
    video = torch.randint(0, 256, (12, 230, 250, 3), generator=g, dtype=torch.uint8)
    span = [1, 3, 5, 7, 9, 11, 0, 2]
    rects = torch.zeros(8, 2, 4, dtype=torch.int32)
    rects[0, 0] = torch.tensor([240, 250, 0, 230]); rects[3, 1] = torch.tensor([0, 30, 200, 230]); rects[7, 0] = torch.tensor([10, 10, 5, 50])   # last: empty
    d, a, m = ops.clip_from_u8(video.cuda(), span, 6, 26, rects.cuda())
    ref = (video[span][:, 6:230, 26:250].double() / 255.).permute(3, 0, 1, 2)
    assert torch.equal(d.cpu(), ref.float()) and torch.equal(a.cpu(), ref.flip(3).float())
    mm = torch.zeros(8, 230, 250); mm[0, 0:230, 240:250] = 1; mm[3, 200:230, 0:30] = 1
    assert torch.equal(m.cpu(), mm[:, 6:230, 26:250])
    with pytest.raises(RuntimeError, match="outside"):
        ops.clip_from_u8(video.cuda(), span, 7, 26, rects.cuda())
    with pytest.raises(RuntimeError, match="frame"):
        ops.clip_from_u8(video.cuda(), [0, 1, 2, 3, 4, 5, 6, 12], 0, 0, None)


# ---------------------------------------------------------------- JHMDB form (datasets/jhmdb_dataloader.py:102-230)
@pytest.mark.parametrize("k", range(fx.N_JHMDB))
def test_oracle_matches_reference_jhmdb_loader(k):
    G = np.load(GOLD)
    frames, masks, label, ann, train = fx.jhmdb_case(k)
    np.random.seed(2000 + k)
    s = oi.get_item_jhmdb(frames, masks.copy(), label, ann, train)
    d, a, m, mc = s["data"].numpy(), s["aug_data"].numpy(), s["loc_msk"].numpy(), s["mask_cls"].numpy()
    assert np.array_equal(d[:, :, ::9, ::7], G["jdata_%d" % k]) and np.array_equal(a[:, :, ::9, ::7], G["jaug_%d" % k])
    assert np.array_equal(np.packbits(m.astype(np.uint8)), G["jmask_%d" % k]) and np.array_equal(mc[0, :, 0, 0], G["jmcls_%d" % k])
    assert np.array_equal(np.array([d.sum(), a.sum(), m.sum(), mc.sum(), float(s["action"][0])]), G["jsums_%d" % k])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(fx.N_JHMDB))
def test_device_pipeline_matches_reference_jhmdb_loader(k):
    from picons_amd import inputpipe as ip
    G = np.load(GOLD)
    frames, masks, label, ann, train = fx.jhmdb_case(k)
    np.random.seed(2000 + k)
    s = ip.get_item_jhmdb(frames, masks.copy(), label, ann, train)
    np.random.seed(2000 + k)
    o = oi.get_item_jhmdb(frames, masks.copy(), label, ann, train)
    for key in ("data", "aug_data", "loc_msk", "mask_cls"):
        got = s[key].cpu().numpy()
        assert got.dtype == np.float32 and np.array_equal(got, o[key].numpy().astype(np.float32)), key
    assert np.array_equal(np.packbits(s["loc_msk"].cpu().numpy().astype(np.uint8)), G["jmask_%d" % k])
    assert np.array_equal(s["mask_cls"].cpu().numpy()[0, :, 0, 0], G["jmcls_%d" % k].astype(np.float32))
    assert float(s["action"][0]) == G["jsums_%d" % k][4]
    if k == 1:
        assert 0 < G["jmcls_%d" % k].sum() < 8            # the sparse case really has frames without truth inside the window


@pytest.mark.gpu
def test_sample_stager_writes_the_minibatch_in_place():
    """StepEngine.sample_stager(): get_item(..., out=...) writes every sample into its place (after cat + randperm shuffle, main_ucf101.py:65-79)
    of the next step's float32 staging and the step reads it there.  Same clips, same draws, through get_item + StepEngine.stage (stack, cat,
    gather, copy into the arena): bit-identical losses and outputs, and the staging holds exactly the staged minibatch."""
    from picons_amd import inputpipe as ip, step as pstep, synthetic
    dev = "cuda:0"
    bs = 4
    args = pstep.default_args(bv=True, n_frames=5, wt_cons=0.1)
    vids = [synthetic.make_decoded_video(300 + i, labeled=i < bs // 2, num_classes=24) for i in range(bs)]
    perm = np.array([2, 0, 3, 1])
    drops = [(np.random.RandomState(5 + q).rand(bs, c) < 0.5).astype(np.float32) * 2 for q, c in enumerate((832, 128, 832, 128))]
    ramp = pstep.exp_rampup(100)(1)

    def collate(samples):
        return {'data': torch.stack([s['data'] for s in samples]), 'aug_data': torch.stack([s['aug_data'] for s in samples]),
                'loc_msk': torch.stack([s['loc_msk'] for s in samples]), 'action': torch.stack([s['action'] for s in samples]),
                'label_vid': torch.tensor([s['label_vid'] for s in samples])}
    eng = pstep.StepEngine(args, bs=bs, hw=224, device=dev)
    np.random.seed(77)
    smp = [ip.get_item(*v, train=True) for v in vids]
    eng.stage(collate(smp[:bs // 2]), collate(smp[bs // 2:]), perm, drops)
    eng.forward_backward(1, ramp)
    a = eng.read_scalars()
    out_a = [t.clone() for t in eng.outputs()]
    want = torch.stack([s['data'] for s in smp])[torch.as_tensor(perm)]

    st = eng.sample_stager()
    np.random.seed(77)
    st.prepare(0, lambda i, out: ip.get_item(*vids[i], train=True, out=out, ndhwc4=True), bs // 2, perm, drops)
    st.commit(0)
    eng.forward_backward(1, ramp)
    st.release(0)
    b = eng.read_scalars()
    out_b = eng.outputs()
    torch.cuda.synchronize()
    x = st.x[0]                                                  # [2 bs][T][H][W][4]: the clips, then the flipped clips, RGB + a zero
    assert torch.equal(x[:bs, ..., :3], want.permute(0, 2, 3, 4, 1)) and torch.equal(x[bs:, ..., :3], torch.flip(want, [4]).permute(0, 2, 3, 4, 1))
    assert torch.all(x[..., 3] == 0)
    for k in ("total", "loc", "cls", "cons"):
        assert a[k] == b[k], (k, a[k], b[k])
    for x, y in zip(out_a, out_b):
        assert torch.equal(x, y)
    assert torch.equal(eng.labels_host, torch.tensor([smp[i]['label_vid'] for i in perm], dtype=torch.int32))
    # back on the arena path (stage() undoes the re-pointing): the first result again, and the planar out= form of get_item
    eng.stage(collate(smp[:bs // 2]), collate(smp[bs // 2:]), perm, drops)
    eng.forward_backward(1, ramp)
    c = eng.read_scalars()
    assert c["total"] == a["total"]
    np.random.seed(77)
    d0, a0, m0 = torch.empty(3, 8, 224, 224, device=dev), torch.empty(3, 8, 224, 224, device=dev), torch.empty(1, 8, 224, 224, device=dev)
    s0 = ip.get_item(*vids[0], train=True, out=(d0, a0, m0))
    assert torch.equal(d0, smp[0]["data"]) and torch.equal(a0, smp[0]["aug_data"]) and torch.equal(m0, smp[0]["loc_msk"]) and s0["data"] is d0
