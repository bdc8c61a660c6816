"""Run-time switches of the planner and the step engine, in two classes.

PRODUCT switches choose between configurations that give the same results up to fp32 rounding and that the parity suite has been run green
under -- the A/B switch of a design decision that stands.  They are read from the environment: `get(name, default)`.

EXPERIMENT switches select what the parity suite REJECTED (arithmetic that misses a bar: F(4x4, 3x3) in front of EM routing, the trunk's forward
convs on the bf16 split) or what was measured and gave nothing (BatchNorm finalize folded into the apply kernels, later skip-conv start,
priorities, deferral, grouped weight-gradient launches, lane binding orders, the probes that restrict the split to some launches).  One environment
variable must not be enough to run the product on them: `exp(name, default, override)` honours the environment ONLY with PICONS_DIAG_LIB=1 (the
diagnostic library build, as the kernel ablations since round 3); a test or a probe tool may pass them explicitly (Plan(exp={...}) /
StepEngine(exp={...})).  tests/test_capi_cpu.py::test_experiment_switches_do_not_reach_the_product_plan builds the plan with every one of them set
and requires it to be identical to the default plan."""
import os

EXPERIMENTS = (
    "PICONS_WINO4_TRUNK_FWD", "PICONS_WINO4_MIN_TILES", "PICONS_SPLIT_TRUNK_FWD",       # fail step-level parity bars
    "PICONS_BN_FUSED", "PICONS_SKIP_FWD_AFTER", "PICONS_PRIO", "PICONS_DEFER_SIDE",        # measured: no gain or slower (docs/MEASUREMENTS.md)
    "PICONS_WGRAD_MULTI", "PICONS_WGRAD_MULTI_TAIL", "PICONS_BIND_LANES", "PICONS_BIND_ORDER", "PICONS_BUCKETS_JOINED", "PICONS_STAGE_STREAM_PRIO",
    "PICONS_SPLIT_ONLY_SPECTRAL", "PICONS_SPLIT_LISTS", "PICONS_SPLIT_CI_MIN", "PICONS_SPLIT_CI_MAX", "PICONS_SPLIT_ROWS_MIN", "PICONS_SPLIT_ROWS_MAX",   # probes
)


def diag():
    return os.environ.get("PICONS_DIAG_LIB", "0") not in ("", "0")


def get(name, default):
    """A product switch: the environment's value, else the default."""
    assert name not in EXPERIMENTS, name
    return os.environ.get(name, default)


def exp(name, default, override=None):
    """An experiment switch: `override[name]` if the caller passed one, the environment only under PICONS_DIAG_LIB=1, else the default."""
    assert name in EXPERIMENTS, name
    if override and name in override:
        return str(override[name])
    return os.environ.get(name, default) if diag() else default
