"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/picons.h
declares (no compute calls without a GPU); product ops refuse CPU tensors."""
import os
import re

import pytest
import torch

import __graft_entry__ as ge
from picons_amd import capi, desc, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(capi.LIB_PATH):
        ge.build()
    return capi.lib()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "picons.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    for n in names:
        assert hasattr(built, n), "libpicons.so does not export %s" % n
    assert sorted(n for n in names) == sorted(capi.EXPORTS), set(names) ^ set(capi.EXPORTS)
    assert built.pc_version() == capi.ABI_VERSION            # the library and the Python mirror of include/picons.h agree (capi.lib() refuses otherwise)
    assert ("#define PC_VERSION %d" % capi.ABI_VERSION) in hdr


def test_product_library_holds_no_wrong_result_diagnostics(built):
    """the ablation / stamp variants of the GEMM kernels (some WRONG by design) and their switches are compiled only into
    libpicons_diag.so (make diag); the shipped library does not even contain the variable names, and setting one without the diagnostic
    library is refused by the loader instead of being ignored."""
    import subprocess
    import sys
    blob = open(capi.LIB_PATH, "rb").read()
    for word in (b"ABLATE", b"PICONS_WINO_VARIANT"):
        assert word not in blob, "%s found in the product library" % word.decode()
    code = "import picons_amd; from picons_amd import capi; capi.lib()"
    env = dict(os.environ, PICONS_WGRAD_ABLATE="1", PYTHONPATH=ROOT)
    env.pop("PICONS_DIAG_LIB", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.returncode != 0 and "diagnostic build" in p.stderr


def test_experiment_switches_do_not_reach_the_product_plan(built, monkeypatch):
    """the switches that select arithmetic the parity suite REJECTED (F(4x4, 3x3) in front of EM routing, the trunk's forward convs
    on the bf16 split) or configurations measured to give nothing are honoured only with PICONS_DIAG_LIB=1 or when passed explicitly
    (picons_amd/switches.py).  A default plan built with every one of them set in the environment is identical, op for op, to one built
    without; passed explicitly they do change the plan."""
    from picons_amd import step as pstep, switches as sw
    from picons_amd.plan import Plan

    def build(exp=None):
        p = Plan(24, 112, n=1, groups=2, lanes=4, early_adam=True, exp=exp)
        p.build_forward(); p.build_loss(pstep.default_args(bv=True, n_frames=5)); p.build_backward(); p.build_adam(); p.finalize()
        return [(n, [(op[0], tuple(op[1]), tuple(op[2]), tuple(map(str, op[3])), tuple(op[4]), op[5]) for op in lst]) for n, lst in p.lists.items()]
    monkeypatch.delenv("PICONS_DIAG_LIB", raising=False)
    base = build()
    values = {"PICONS_WINO4_MIN_TILES": "49", "PICONS_SKIP_FWD_AFTER": "Mixed_3c", "PICONS_SPLIT_LISTS": "fwd", "PICONS_SPLIT_CI_MAX": "64",
              "PICONS_SPLIT_ROWS_MAX": "1000", "PICONS_BIND_ORDER": "0,3,2,1"}
    for name in sw.EXPERIMENTS:
        monkeypatch.setenv(name, values.get(name, "1"))
    assert build() == base, "an experiment switch in the environment changed the product plan"
    assert sw.exp("PICONS_WINO4_TRUNK_FWD", "0") == "0" and sw.exp("PICONS_WINO4_TRUNK_FWD", "0", {"PICONS_WINO4_TRUNK_FWD": "1"}) == "1"
    for name in sw.EXPERIMENTS:
        monkeypatch.delenv(name)
    assert build({"PICONS_BN_FUSED": "1"}) != base
    with pytest.raises(AssertionError):
        sw.get("PICONS_BN_FUSED", "0")                  # an experiment cannot be read as a product switch


def test_struct_layouts_match_header(built):
    import ctypes as C
    assert C.sizeof(capi.ConvDesc) == 48 * 4
    assert C.sizeof(capi.WgradDesc) == 42 * 4
    assert C.sizeof(capi.PoolDesc) == 19 * 4
    assert C.sizeof(capi.LossDesc) == 16 * 4
    assert capi.OP_DTYPE.itemsize == 360 and capi.OP_DTYPE.fields["p"][1] == 232
    assert len(desc.flatten(desc.conv_fwd(1, (1, 2, 2), 4, 4, 4, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 2)), desc.CONV_FIELDS)) == 48


def test_no_cpu_fallback(built):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv_fwd(desc.conv_fwd(1, (1, 2, 2), 4, 4, 4, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 2)),
                     torch.zeros(1, 1, 2, 2, 4), torch.zeros(4, 1, 4), torch.zeros(1, 1, 2, 2, 4))


def test_bad_descriptor_is_reported_without_gpu(built):
    import ctypes as C
    d = ops.conv_desc(desc.conv_fwd(1, (1, 2, 2), 3, 3, 4, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 2)))
    rc = built.pc_conv_fwd(C.byref(d), C.c_void_p(16), C.c_void_p(16), None, None, C.c_void_p(16), None, None)
    assert rc == -1 and b"multiples of 4" in built.pc_last_error()


def test_tools_and_dropin_scripts_compile():
    """Every script under tools/ and dropin/ at least parses (they only run on a GPU box or in the authoring container)."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "pi-consistency-activity-detection_amd", "dropin", "**", "*.py"), recursive=True) + \
        [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 15
    for f in files:
        compile(open(f).read(), f, "exec")


def test_host_side_of_the_abi_under_asan_ubsan():
    """the host side of the C-ABI under AddressSanitizer + UBSan in the CPU container (GPU ASan needs XNACK, which
    the pool does not offer).  `make asan` builds libpicons_asan.so (host code instrumented, device code as usual) and
    tests/capi_host_driver.cpp, which walks every argument-checking path and the host arithmetic (workspace sizes, the
    cv2.resize tables written into caller memory) without a GPU; a sanitizer report or a failed expectation fails the run."""
    import subprocess
    csrc = os.path.join(ROOT, "pi-consistency-activity-detection_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-j8", "asan"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(csrc, "asan", "capi_host_driver")], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_tensors_beyond_the_lds_dma_offset_range_are_refused_without_gpu(built):
    """The gather kernels address their operands with 32-bit byte offsets from a wave-uniform base (buffer-resource LDS-DMA): a tensor of
    4 GiB or more is refused by the launcher (host check, before any HIP call) instead of wrapping.  bs = 8 is 0.8 GB at most."""
    import ctypes as C
    # conv: 64 samples x 16 frames x 224^2 x 128 channels of fp32 = 26 GB of input
    d = ops.conv_desc(desc.trim_conv(desc.conv_fwd(64, (16, 224, 224), 128, 128, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (16, 224, 224), groups=2)))
    rc = built.pc_conv_fwd(C.byref(d), C.c_void_p(16), C.c_void_p(16), None, None, C.c_void_p(16), None, None)
    assert rc != 0 and b"4 GiB" in built.pc_last_error()
    # weight gradient of the same layer
    wd = ops._fill_struct(capi.WgradDesc(), desc.trim_wgrad(desc.wgrad(64, (16, 224, 224), 128, 128, (16, 224, 224), 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1))))
    rc = built.pc_conv_wgrad(C.byref(wd), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None)
    assert rc != 0 and (b"4 GiB" in built.pc_last_error() or b"out of range" in built.pc_last_error())
    # Winograd: one frame of 4 GiB
    w = ops.wino_desc(1, 1, 8192, 8192, 32, 32, 64, 64, 3)
    out = (C.c_double * 3)()
    assert built.pc_wino_work(C.byref(w), out) != 0 and b"plane too large" in built.pc_last_error()
