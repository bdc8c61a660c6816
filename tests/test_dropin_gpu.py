"""GPU: the nn.Module drop-in (CapsNet / InceptionI3d / utils.helpers) behind the reference's surface:
forward, autograd backward with two forward calls per step and gradient accumulation, eval mode,
state_dict round trip - all against the CPU oracle on the same inputs."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import caps as ocaps, i3d as oi3d, step as ostep
from picons_amd import model as pmodel, spec, synthetic

pytestmark = pytest.mark.gpu
HW = 112


def _oracle_params(ncls=24):
    return ostep.as_torch_params(synthetic.init_state(47, ncls))


def test_state_dict_surface():
    m = pmodel.CapsNet(pt_path=None, hw=HW, init="conditioned")
    sd = m.state_dict()
    assert list(sd.keys()) == spec.state_dict_keys(24) and len(sd) == 293
    ref = synthetic.init_state(47, 24)
    for k in ("conv1.Mixed_4f.b3b.conv3d.weight", "primary_caps.pose.weight", "upsample4.weight", "conv_caps.weights"):
        assert tuple(sd[k].shape) == ref[k].shape and np.array_equal(sd[k].cpu().numpy(), ref[k])
    assert sum(p.numel() for p in m.parameters()) == 48003705
    m2 = pmodel.CapsNet(pt_path=None, hw=HW, seed=3, init="conditioned")
    m2.load_state_dict(sd)
    assert torch.equal(m2.state_dict()["smooth.weight"], sd["smooth.weight"])


def test_default_init_follows_the_reference_and_missing_trunk_raises(monkeypatch, tmp_path):
    """CapsNet() starts from what constructing the reference's module leaves in its parameters (capsules_ucf101.py:36,39,
    97-103,359-374; pytorch_i3d.py:69-80), not from the parity-test initialiser; a missing rgb_charades.pt raises like the
    reference's torch.load (:344) unless PICONS_SYNTHETIC says the run is synthetic."""
    m = pmodel.CapsNet(pt_path=None, hw=HW)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    assert abs(float(sd["primary_caps.pose.weight"].std()) - 0.1) < 2e-3 and abs(float(sd["primary_caps.a.weight"].std()) - 0.1) < 2e-3
    assert abs(float(sd["conv_caps.weights"].std()) - 1.0) < 2e-2 and abs(float(sd["conv_caps.beta_u"].std()) - 1.0) < 0.15
    for k in ("upsample1.weight", "upsample4.weight", "smooth.weight"):
        assert abs(float(sd[k].std()) - 0.02) < 2e-3, k
    for k, v in sd.items():
        if k.endswith(".bn.weight"):
            assert torch.all(v == 1), k
        if k.endswith(".bn.bias"):
            assert torch.all(v == 0), k
    w = sd["conv112.weight"]
    bound = 1.0 / (64 * 27) ** 0.5                        # PyTorch default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), bias too
    assert float(w.abs().max()) <= bound and float(w.abs().max()) > 0.95 * bound
    assert float(sd["conv112.bias"].abs().max()) <= bound
    assert float(sd["upsample4.bias"].abs().max()) <= 1.0 / (128 * 27) ** 0.5        # ConvTranspose: fan_in from weight.shape[1]
    monkeypatch.delenv("PICONS_SYNTHETIC", raising=False)
    with pytest.raises(FileNotFoundError):
        pmodel.CapsNet(pt_path=str(tmp_path / "no_such_rgb_charades.pt"), hw=HW)
    monkeypatch.setenv("PICONS_SYNTHETIC", "1")
    pmodel.CapsNet(pt_path=str(tmp_path / "no_such_rgb_charades.pt"), hw=HW)       # warns, keeps the random trunk


def test_forward_without_backward_does_not_leak_arena_slots():
    """A training-mode forward under no_grad takes no slot; a forward whose graph is dropped without backward gives its
    slot back (the autograd ctx's guard), so repeated calls keep using ONE arena instead of allocating a new one each time."""
    m = pmodel.CapsNet(pt_path=None, hw=HW, init="conditioned").cuda()
    m.train(mode=True); m.training = True
    x = torch.rand(2, 3, 8, HW, HW, device="cuda")
    act, lab = torch.zeros(2, 1, device="cuda"), torch.tensor([1, 0], device="cuda")
    with torch.no_grad():
        for _ in range(3):
            m(x, act, lab, 1, 11)
    assert len(m._slots[(2, True)]) == 1 and not m._slots[(2, True)][0].busy
    for _ in range(3):
        out, pred, _f = m(x, act, lab, 1, 11)
        assert m._slots[(2, True)][0].busy
        del out, pred, _f                                  # graph dropped, no backward
    assert len(m._slots[(2, True)]) == 1 and not m._slots[(2, True)][0].busy
    a, _p, _f = m(x, act, lab, 1, 11)
    b, _p2, _f2 = m(x, act, lab, 1, 11)                    # the reference's two passes: two live graphs -> two slots
    assert len(m._slots[(2, True)]) == 2
    (a.sum() + b.sum()).backward()
    assert not any(s.busy for s in m._slots[(2, True)])


def test_module_forward_backward_two_passes_vs_oracle():
    torch.manual_seed(0)
    m = pmodel.CapsNet(pt_path=None, hw=HW, init="conditioned").cuda()
    m.train(mode=True); m.training = True
    lab, unl, perm, _ = synthetic.make_step_inputs(2, hw=HW)
    data = torch.cat([torch.from_numpy(lab["data"]), torch.from_numpy(unl["data"])]).float()
    aug = torch.cat([torch.from_numpy(lab["aug_data"]), torch.from_numpy(unl["aug_data"])]).float()
    action = torch.cat([torch.from_numpy(lab["action"]), torch.from_numpy(unl["action"])])
    labels = torch.tensor([1, 0])
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, eps=1e-6)
    opt.zero_grad()
    # script the module's dropout draws so the oracle can replay them
    draws = []
    orig_rand = torch.rand

    def rec_rand(*a, **k):
        r = orig_rand(*a, **k)
        draws.append(r.detach().cpu())
        return r
    torch.rand = rec_rand
    try:
        out, pred, feat = m(data.cuda(), action.cuda(), labels.cuda(), 1, 11)
        flip, _, _ = m(aug.cuda(), action.cuda(), labels.cuda(), 1, 11)
    finally:
        torch.rand = orig_rand
    d = [(x < 0.5).float() * 2 for x in draws]
    loss = (out ** 2).mean() + (torch.flip(flip, [4]) - out).pow(2).mean() + pred.sum()
    loss.backward()
    P = _oracle_params()
    o_out, o_pred, o_feat = ocaps.capsnet_forward(P, data, action, labels, 1, 11, True, d[0].view(2, 832), d[1].view(2, 128))
    o_flip, _, _ = ocaps.capsnet_forward(P, aug, action, labels, 1, 11, True, d[2].view(2, 832), d[3].view(2, 128))
    ((o_out ** 2).mean() + (torch.flip(o_flip, [4]) - o_out).pow(2).mean() + o_pred.sum()).backward()
    assert (out.cpu() - o_out).abs().max().item() <= 1e-3 and (pred.cpu() - o_pred).abs().max().item() <= 1e-3
    assert (flip.cpu() - o_flip).abs().max().item() <= 1e-3 and (feat.cpu() - o_feat).abs().max().item() <= 5e-3
    num = den = 0.0
    for (k, p) in m.named_parameters():
        g, r = p.grad.cpu().double(), P[k].grad.double()
        num += (g - r).norm().item() ** 2; den += r.norm().item() ** 2
    assert (num / den) ** 0.5 <= 3e-2, (num / den) ** 0.5
    before = m.state_dict()["smooth.weight"].clone()
    opt.step()
    assert not torch.equal(before, m.state_dict()["smooth.weight"])
    # zero_grad + a second backward starts from zero (set_to_none semantics)
    opt.zero_grad()
    out2, pred2, _ = m(data.cuda(), action.cuda(), labels.cuda(), 1, 11)
    out2.mean().backward()
    assert all(p.grad is not None for p in m.parameters())
    assert m.state_dict()["conv1.Conv3d_1a_7x7.bn.num_batches_tracked"].item() == 3


def test_eval_forward_and_trunk_vs_oracle():
    m = pmodel.CapsNet(pt_path=None, hw=HW, init="conditioned").cuda()
    m.eval(); m.training = False
    lab, unl, _, _ = synthetic.make_step_inputs(2, hw=HW)
    data = torch.cat([torch.from_numpy(lab["data"]), torch.from_numpy(unl["data"])]).float()
    action = torch.zeros(2, 1)
    out, pred, _ = m(data.cuda(), action.cuda(), torch.zeros(2).cuda(), 0, 0)
    P = _oracle_params()
    with torch.no_grad():
        o_out, o_pred, _ = ocaps.capsnet_forward(P, data, action, torch.zeros(2), 0, 0, False, None, None)
    assert (out.cpu() - o_out).abs().max().item() <= 1e-3 and (pred.cpu() - o_pred).abs().max().item() <= 1e-3
    tr = pmodel.InceptionI3d(157, in_channels=3, final_endpoint='Mixed_4f', hw=HW)
    tr.load_state_dict({k[len("conv1."):]: v for k, v in synthetic.init_state(47, 24).items() if k.startswith("conv1.")})
    tr.eval()
    x, o56, o112 = tr(data.cuda())
    with torch.no_grad():
        rx, r56, r112 = oi3d.trunk(P, data, False)
    for a, b in ((x, rx), (o56, r56), (o112, r112)):
        assert tuple(a.shape) == tuple(b.shape) and (a.cpu() - b).abs().max().item() <= 1e-3 * max(1.0, b.abs().max().item())


def test_helpers_dropin_matches_reference_golden(golden_dir):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi-consistency-activity-detection_amd", "dropin"))
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "utils" or k.startswith("utils.")}
    try:
        from utils.helpers import measure_pixelwise_var_v2, measure_pixelwise_gradient
        M = np.load(os.path.join(golden_dir, "masks.npz"))
        g = np.random.default_rng(23)
        pred = g.normal(0, 2, (2, 1, 8, 224, 224)).astype(np.float32)
        flip = (pred[:, :, ::-1] * 0.7 + g.normal(0, 1, pred.shape)).astype(np.float32)
        v = measure_pixelwise_var_v2(torch.from_numpy(pred).cuda(), torch.from_numpy(flip).cuda(), frames_cnt=5)
        assert list(v.shape) == [2, 1, 8, 224, 224] and np.abs(v.cpu().numpy()[..., ::7, ::7] - M["var5_sample"]).max() <= 2e-5
        gm = measure_pixelwise_gradient(torch.from_numpy(pred).cuda(), 0.2, 0.85)
        assert list(gm.shape) == [2, 8, 224, 224] and np.abs(gm.cpu().numpy()[..., ::7, ::7] - M["grad_thr_sample"]).max() <= 2e-5
        with pytest.raises(UnboundLocalError):
            measure_pixelwise_var_v2(torch.from_numpy(pred).cuda(), torch.from_numpy(flip).cuda(), frames_cnt=4)
    finally:
        sys.path.pop(0)
        for k in list(sys.modules):
            if k == "utils" or k.startswith("utils."):
                del sys.modules[k]
        sys.modules.update(saved)


def test_pretrained_trunk_and_reference_checkpoint_loading(tmp_path):
    """§8f rank 4: `rgb_charades.pt` (a full 400-class I3D state_dict without the 'conv1.' prefix, capsules_ucf101.py:343-352)
    fills exactly the trunk up to Mixed_4f and ignores the layers the model does not have; a checkpoint written the way the
    reference writes them (`torch.save(model.state_dict())`, main_ucf101.py:442,451) comes back through load_previous_weights."""
    ref = synthetic.init_state(3, 24)
    g = torch.Generator().manual_seed(9)
    pre = {k[len("conv1."):]: torch.from_numpy(np.asarray(v).copy()) for k, v in ref.items() if k.startswith("conv1.")}
    for k in list(pre):
        if pre[k].dtype == torch.float32:
            pre[k] = torch.randn(pre[k].shape, generator=g) * 0.05
    pre["Mixed_5b.b0.conv3d.weight"] = torch.randn(256, 832, 1, 1, 1, generator=g)        # layers beyond Mixed_4f / the logits head
    pre["logits.conv3d.weight"] = torch.randn(400, 1024, 1, 1, 1, generator=g)
    pre["logits.conv3d.bias"] = torch.randn(400, generator=g)
    path = str(tmp_path / "rgb_charades.pt")
    torch.save(pre, path)
    m = pmodel.CapsNet(pt_path=path, hw=HW, init="conditioned")
    base = pmodel.CapsNet(pt_path=None, hw=HW, init="conditioned").state_dict()
    sd = m.state_dict()
    for k in sd:
        if k.startswith("conv1."):
            assert torch.equal(sd[k].cpu(), pre[k[len("conv1."):]].to(sd[k].dtype)), k
        else:
            assert torch.equal(sd[k].cpu(), base[k].cpu()), k                               # head / decoder keep their initialisation
    ck = str(tmp_path / "best_model_train_loss_1.pth")
    torch.save(sd, ck)
    m2 = pmodel.CapsNet(pt_path=None, hw=HW, seed=5)
    m2.load_previous_weights(ck)
    sd2 = m2.state_dict()
    assert list(sd2) == list(sd) and all(torch.equal(sd2[k].cpu(), sd[k].cpu()) for k in sd)


@pytest.mark.parametrize("script,ncls,env", [("main_ucf101.py", 24, {"PICONS_SYNTHETIC": "u8"}), ("main_ucf101.py", 24, {"PICONS_SYNTHETIC": "1"}),
                                             ("main_ucf101.py", 24, {"PICONS_SYNTHETIC": "u8", "PICONS_FUSED": "0"}),
                                             ("main_jhmdb.py", 21, {"PICONS_SYNTHETIC": "1"})])
def test_dropin_main_runs_an_epoch(tmp_path, script, ncls, env):
    """dropin/main_ucf101.py end to end with the reference's flags: synthetic decoded uint8 videos through the device input
    pipeline (u8) or float64 minibatches (1), fused step engine or nn.Module + autograd + torch Adam (PICONS_FUSED=0); one
    epoch of two steps, validation, the two checkpoints of the reference's policy."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PICONS_STEPS="2", **env)
    p = subprocess.run([sys.executable, os.path.join(root, "pi-consistency-activity-detection_amd", "dropin", script), "--bs", "4", "--epochs", "1",
                        "--bv", "--n_frames", "5", "--exp_id", "t"], cwd=str(tmp_path), env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "Training time" in p.stdout and "[VAL] epoch-1" in p.stdout
    runs = os.listdir(str(tmp_path / "train_log_wts" / "t"))
    assert len(runs) == 1
    files = sorted(os.listdir(str(tmp_path / "train_log_wts" / "t" / runs[0])))
    assert files == ["best_model_train_loss_1.pth", "best_model_val_loss_1.pth"]
    sd = torch.load(str(tmp_path / "train_log_wts" / "t" / runs[0] / files[0]), map_location="cpu")
    assert list(sd.keys()) == spec.state_dict_keys(ncls) and all(torch.isfinite(v.float()).all() for v in sd.values())
