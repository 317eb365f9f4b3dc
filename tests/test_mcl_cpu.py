"""CPU-only host-logic tests of carle_amd.mcl: module/parameter naming, RNG-order-identical
initialisation and strict loading of reference-format state_dicts (no compute calls)."""
import numpy as np
import torch

import golden_util as G


def _build(size, n):
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D
    env = CARLE(width=size, height=size, instances=n, device="cpu")
    rnd = RND2D(env)
    ae = AE2D(rnd)
    return env, rnd, ae


def test_seeded_construction_matches_reference_weights():
    """Same layer types in the same order => same RNG consumption => identical initial weights
    as the reference under torch.manual_seed(0) (predictor before random net, mcl.py:248-249)."""
    z = G.nets(64)
    torch.manual_seed(0)
    env, rnd, ae = _build(64, 16)
    for prefix, mod in (("rnd_sd", rnd), ("ae_sd", ae)):
        ref = G.load_sd(z, prefix)
        sd = mod.state_dict()
        assert list(sd.keys()) == list(ref.keys())
        for k in ref:
            assert np.array_equal(sd[k].numpy(), ref[k]), k


def test_shipped_format_state_dicts_load_strictly():
    z = G.nets(256)
    env, rnd, ae = _build(256, 4)
    for prefix, mod in (("rnd_sd", rnd), ("ae_sd", ae)):
        ref = {k: torch.from_numpy(v) for k, v in G.load_sd(z, prefix).items()}
        mod.load_state_dict(ref, strict=True)
        for k, v in mod.state_dict().items():
            assert torch.equal(v.cpu(), ref[k]), k


def test_wrapper_surface():
    env, rnd, ae = _build(64, 2)
    assert rnd.my_name == "RND2D" and ae.my_name == "AE2D"
    assert rnd.inner_env is env and ae.inner_env is env and ae.env is rnd and rnd.env is env
    assert (rnd.rnd_dim, rnd.batch_size, rnd.buffer_length, rnd.updates) == (16, 64, 0, 0)
    assert rnd.learning_rate == 6e-2 and ae.learning_rate == 9e-2
    assert ae.optimizer.param_groups[0]["lr"] == 6e-2          # the reference's quirk, SURVEY a15
    assert rnd.reward_scale == 1.0
    assert all(p.requires_grad for p in rnd.predictor.parameters())
    assert not any(p.requires_grad for p in rnd.random_network.parameters())
    assert not hasattr(ae, "random_network")
    ae.rules_from_string("B36/S23")
    assert env.birth == [3, 6] and ae.birth == [3, 6] and ae.survive == [2, 3]
    assert sum(p.numel() for p in rnd.predictor.parameters()) == 1117
    assert sum(p.numel() for p in rnd.random_network.parameters()) == 1079
    assert sum(p.numel() for p in ae.predictor.parameters()) == 164


def test_param_blob_tracks_updates():
    from carle_amd.mcl import _ParamBlob
    lin = torch.nn.Linear(3, 2)
    blob = _ParamBlob(lin)
    a = blob.get(torch.device("cpu"))
    assert blob.get(torch.device("cpu")) is a                 # cached
    with torch.no_grad():
        lin.weight.add_(1.0)
    b = blob.get(torch.device("cpu"))
    assert b is not a and torch.equal(b[:6], lin.weight.detach().reshape(-1))


def test_train_warns_under_sharding_without_a_process_group(monkeypatch):
    """harness.train says so when WORLD_SIZE > 1 and no process group is up (per-rank predictors, no gradient
    all-reduce); with a group it enables carle_amd.dist.data_parallel instead (tests/test_gpu_stats.py)."""
    import warnings
    from carle_amd import harness
    monkeypatch.setenv("WORLD_SIZE", "2")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        harness._data_parallel_or_warn(None)
    assert any("every rank trains its own" in str(w.message) for w in rec)
    monkeypatch.setenv("WORLD_SIZE", "1")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        harness._data_parallel_or_warn(None)
    assert not rec
