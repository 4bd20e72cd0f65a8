"""CPU: host logic added in round 6 -- the `unopose_amd.ops` package (one module per operator family, switches forwarded to `ops._state`),
the derived-weight cache key, the shared-reference synthetic batch of bench.py's `ref_cached` leg and the process-wide stream pools' keys.
No GPU work: the library is only loaded for symbol checks elsewhere (tests/test_abi.py)."""
import importlib

import pytest
import torch


def test_switches_are_forwarded_to_the_state_module_and_seen_by_the_families():
    from unopose_amd import ops
    from unopose_amd.ops import _state, dense

    assert ops.USE_LN_FOLD is True and _state.USE_LN_FOLD is True
    ops.USE_LN_FOLD = False  # what scripts/ubench/bench_ab.py and the tests do
    try:
        assert _state.USE_LN_FOLD is False and ops.USE_LN_FOLD is False
        assert dense.st is _state  # the family modules read the switch at call time through this module
        assert "USE_LN_FOLD" not in vars(ops)  # never shadowed by a package attribute
    finally:
        ops.USE_LN_FOLD = True
    assert _state.USE_LN_FOLD is True
    with pytest.raises(AttributeError):
        ops.NO_SUCH_SWITCH  # noqa: B018
    with pytest.raises(AttributeError):
        del ops.USE_LN_FOLD
    # the mode flag follows the context manager wherever it is read from
    assert not ops.is_differentiable() and ops._DIFF is False
    with ops.differentiable():
        assert ops.is_differentiable() and ops._DIFF is True and _state._DIFF is True
        with ops.differentiable(False):
            assert not ops.is_differentiable()
        assert ops.is_differentiable()
    assert not ops.is_differentiable()


def test_every_switch_has_one_home_and_every_family_exports_through_the_package():
    from unopose_amd import ops
    from unopose_amd.ops import _state

    declared = {k for k in vars(_state) if k.isupper() or k == "_DIFF"}
    assert declared == set(ops._SWITCHES), declared ^ set(ops._SWITCHES)
    for fam in ("common", "dense", "attention", "geometry", "sampling", "pose", "train"):
        m = importlib.import_module("unopose_amd.ops." + fam)
        public = [k for k, v in vars(m).items() if callable(v) and getattr(v, "__module__", None) == m.__name__ and not k.startswith("__")]
        assert public, fam
        for k in public:
            assert getattr(ops, k) is getattr(m, k), (fam, k)  # `ops.linear(...)`, `ops._lin(...)` keep working as before the split
        assert not any(k in declared for k in vars(m)), fam  # no family module keeps a private copy of a switch


def test_params_key_follows_an_in_place_edit_of_any_tensor_of_the_module():
    from unopose_amd.ops.common import _params_key

    lin = torch.nn.Linear(8, 4)
    bn = torch.nn.BatchNorm1d(4)
    k0, b0 = _params_key(lin), _params_key(bn, "tag")
    assert _params_key(lin) == k0
    with torch.no_grad():
        lin.bias.add_(1.0)  # a bias alone
    k1 = _params_key(lin)
    assert k1 != k0
    with torch.no_grad():
        lin.weight.mul_(2.0)
    assert _params_key(lin) != k1
    with torch.no_grad():
        bn.running_var.add_(0.5)  # a buffer
    assert _params_key(bn, "tag") != b0 and _params_key(bn, "tag")[-1] == "tag"


def test_shared_reference_batch_is_consistent_geometry():
    """bench.py's `ref_cached` leg: groups of `per_ref` queries look at ONE reference view; every query is that view's cloud under its own pose
    (p_query = R p_ref + t with the returned ground truth), its pixels a subset of the reference's."""
    from unopose_amd.synthetic import make_shared_reference_batch

    ep, keys, R, t = make_shared_reference_batch(8, per_ref=4, nq=128, nt=300, S=56, seed=3, noise=0.0)
    assert len(keys) == 8 and len(set(keys)) == 2 and keys[0] == keys[3] != keys[4]
    assert torch.equal(ep["tem1_pts"][0], ep["tem1_pts"][3]) and not torch.equal(ep["tem1_pts"][0], ep["tem1_pts"][4])
    assert torch.equal(ep["tem1_rgb"][1], ep["tem1_rgb"][2]) and torch.equal(ep["rgb"][1], ep["tem1_rgb"][1])
    for b in range(8):
        ref, q = ep["tem1_pts"][b], ep["pts"][b]
        moved = ref @ R[b].T + t[b]
        # every query point is one of the re-posed reference points (no cdist: its |a|^2 + |b|^2 - 2 a.b form cancels at this scale)
        d = (q[:, None, :] - moved[None, :, :]).norm(dim=-1).min(dim=1)[0]
        assert d.max().item() < 1e-5, (b, d.max().item())
        assert set(ep["rgb_choose"][b].tolist()) <= set(ep["tem1_choose"][b].tolist())
    assert (R @ R.transpose(1, 2) - torch.eye(3)).abs().max().item() < 1e-5


def test_package_import_defaults_the_hardware_queue_count_without_overriding_the_user():
    import os
    import subprocess
    import sys

    code = "import os, unopose_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = root
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "16"
    env["GPU_MAX_HW_QUEUES"] = "2"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "2"
