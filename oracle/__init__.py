"""CPU oracle for the semi-supervised train step — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-fp32 / numpy restatement of the reference's algorithm for the hot path
(SURVEY.md §8a rows a1-a17).  Every function cites the reference file:line it follows.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product path (`pi-consistency-activity-detection_amd/`) never does and fails loudly when
the HIP library is missing.

Parity pin: the reference has no tests, golden vectors or fixtures of its own, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, imported on CPU in the authoring
container by `tools/make_goldens.py` (shims listed there); the resulting vectors are committed
under `tests/golden/` and checked by `tests/test_oracle_golden.py`.
The 21-class JHMDB model file is absent from the reference - parity for that head
is pinned only through the same classes at C=21.
"""
