"""CPU: host orchestration of freud_amd.train_sae.train() against the reference's own train()
(golden fixture trainloop_*.npz produced by tests/golden/make_golden.py), with the oracle-backed
engine stand-in from tests/fake_engine.py in place of the HIP engine.  Pins: RNG consumption /
batch order, LR schedule timing, logging tags, checkpoint file names, keys and contents, resume."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from freud_amd.loader import write_shards
from freud_amd.train_sae import train, lr_at
from tests.fake_engine import OracleEngine


def _setup(tmp_path, golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    meta = json.loads(str(z["meta"]))
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, meta["layer"], z["shard"], [meta["T"], meta["d"]],
                 [f"/data/audio/file_{i:04d}.flac" for i in range(meta["n_files"])])
    cfg = copy.deepcopy(meta["config"])
    cfg.update(train_folder=folder, val_folder=folder, run_dir=os.path.join(str(tmp_path), "run"), device="cpu")
    return z, meta, cfg


def _scalars(run_dir):
    return [json.loads(l) for l in open(os.path.join(run_dir, "metrics.jsonl"))]


def test_l1_train_loop_matches_reference(tmp_path, golden_dir):
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_l1")
    train(**cfg, engine_factory=OracleEngine)
    ck_dir = os.path.join(cfg["run_dir"], "checkpoints")
    assert sorted(os.listdir(ck_dir)) == meta["checkpoint_files"]        # step{save_every}, epoch ends, final
    got = {(s["tag"], s["step"]): s["value"] for s in _scalars(cfg["run_dir"])}
    for tag, val, step in meta["scalars"]:
        assert (tag, step) in got, (tag, step)
        assert got[(tag, step)] == pytest.approx(val, rel=2e-6, abs=1e-12), (tag, step)
    ck = torch.load(os.path.join(ck_dir, "step7.pth"), map_location="cpu", weights_only=True)   # plain containers only
    assert sorted(ck.keys()) == meta["checkpoint_keys"]
    assert list(ck["model"].keys()) == meta["model_keys"]
    assert ck["step"] == meta["step"] and ck["best_val_loss"] == meta["best_val_loss"]
    hp = dict(meta["hparams"])
    hp.update(train_folder=cfg["train_folder"], val_folder=cfg["val_folder"])
    assert ck["hparams"] == hp
    for k in meta["model_keys"]:
        torch.testing.assert_close(ck["model"][k], torch.tensor(z[f"model__{k}"]), rtol=0, atol=2e-7)
    assert sorted(ck["scheduler"].keys()) == meta["scheduler_keys"]
    ref_groups = meta["opt_param_groups"][0]
    assert set(ck["optimizer"]["param_groups"][0].keys()) == set(ref_groups.keys())
    for k, v in ref_groups.items():
        got_v = ck["optimizer"]["param_groups"][0][k]
        if isinstance(v, float):
            assert got_v == pytest.approx(v, rel=1e-9, abs=1e-15), k
        else:
            assert got_v == v or list(got_v) == list(v), k
    for pid in (0, 1):
        st = ck["optimizer"]["state"][pid]
        assert float(st["step"]) == float(z[f"opt__{pid}__step"])
        np.testing.assert_allclose(st["exp_avg"].numpy(), z[f"opt__{pid}__exp_avg"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(st["exp_avg_sq"].numpy(), z[f"opt__{pid}__exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_checkpoint_consumed_by_reference_key_logic(tmp_path, golden_dir):
    """init_sae_from_checkpoint (src/dataset/activations.py:16-31) reads hparams.activation_size,
    hparams.autoencoder_variant, hparams.autoencoder_config and model: restate that key logic."""
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_l1")
    train(**cfg, engine_factory=OracleEngine)
    ck = torch.load(os.path.join(cfg["run_dir"], "checkpoints", "step4.pth"), map_location="cpu")
    d = ck["hparams"]["activation_size"]
    assert ck["hparams"]["autoencoder_variant"] == "l1"
    n = ck["hparams"]["autoencoder_config"]["n_dict_components"]
    lin = torch.nn.Linear(n, d, bias=False)
    lin.load_state_dict({"weight": ck["model"]["decoder.weight"]})
    assert ck["model"]["encoder_bias"].shape == (n,)


def test_resume_from_checkpoint_continues_identically(tmp_path, golden_dir):
    """with the side file save_checkpoint writes (<run_dir>/resume/<name>: RNG state of the
    epoch's permutation draw + batches consumed) an interrupted run continues on exactly the batches the uninterrupted
    one saw; without it the behaviour is the reference's (model / optimizer / step restored, data order restarts)."""
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_l1")
    train(**cfg, engine_factory=OracleEngine)
    full = torch.load(os.path.join(cfg["run_dir"], "checkpoints", "step7.pth"), map_location="cpu")
    assert os.path.exists(os.path.join(cfg["run_dir"], "resume", "step4.pth"))
    side = torch.load(os.path.join(cfg["run_dir"], "resume", "step4.pth"), map_location="cpu", weights_only=True)
    assert side["step"] == 4 and side["epoch_batches_done"] >= 1
    cfg2 = dict(cfg, run_dir=os.path.join(str(tmp_path), "run2"),
                start_checkpoint=os.path.join(cfg["run_dir"], "checkpoints", "step4.pth"))
    st = train(**cfg2, engine_factory=OracleEngine)
    assert st["step"] == 7
    ck = torch.load(os.path.join(cfg2["run_dir"], "checkpoints", "step7.pth"), map_location="cpu")
    for k, v in full["model"].items():                       # identical trajectory: same batches, same arithmetic
        assert torch.equal(ck["model"][k], v), k
    for pid in (0, 1):
        assert torch.equal(ck["optimizer"]["state"][pid]["exp_avg"], full["optimizer"]["state"][pid]["exp_avg"])
    assert ck["scheduler"]["last_epoch"] == 7 == full["scheduler"]["last_epoch"]

    # reference behaviour when the side file is absent (e.g. a checkpoint written by the reference itself)
    os.remove(os.path.join(cfg["run_dir"], "resume", "step4.pth"))
    cfg3 = dict(cfg2, run_dir=os.path.join(str(tmp_path), "run3"))
    st = train(**cfg3, engine_factory=OracleEngine)
    assert st["step"] == 7
    ck3 = torch.load(os.path.join(cfg3["run_dir"], "checkpoints", "step7.pth"), map_location="cpu")
    assert float(ck3["optimizer"]["state"][0]["step"]) == 7.0
    assert ck3["scheduler"]["_last_lr"][0] == pytest.approx(full["scheduler"]["_last_lr"][0], rel=1e-12)
    assert all(torch.isfinite(v).all() for v in ck3["model"].values())
    assert not all(torch.equal(ck3["model"][k], v) for k, v in full["model"].items())   # different batches were seen


def test_topk_resume_restores_dead_latent_counters(tmp_path, golden_dir):
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_topk")
    st_full = train(**cfg, engine_factory=OracleEngine)
    nfsf_full = st_full["engine"].get_topk_state()
    files = sorted(os.listdir(os.path.join(cfg["run_dir"], "resume")))
    mid = [f for f in files if f != "step7.pth"][0]
    side = torch.load(os.path.join(cfg["run_dir"], "resume", mid), map_location="cpu", weights_only=True)
    assert side["num_frames_since_fired"].dtype == torch.int64 and side["num_frames_since_fired"].numel() == nfsf_full.size
    cfg2 = dict(cfg, run_dir=os.path.join(str(tmp_path), "run2"),
                start_checkpoint=os.path.join(cfg["run_dir"], "checkpoints", mid))
    st = train(**cfg2, engine_factory=OracleEngine)
    assert np.array_equal(st["engine"].get_topk_state(), nfsf_full)          # counters continued, not restarted
    full = torch.load(os.path.join(cfg["run_dir"], "checkpoints", "step7.pth"), map_location="cpu")
    ck = torch.load(os.path.join(cfg2["run_dir"], "checkpoints", "step7.pth"), map_location="cpu")
    for k, v in full["model"].items():
        assert torch.equal(ck["model"][k], v), k


def test_error_conventions(tmp_path, golden_dir):
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_l1")
    with pytest.raises(TypeError):
        train(**{k: v for k, v in cfg.items() if k != "clip_thresh"}, engine_factory=OracleEngine)
    with pytest.raises(TypeError):
        train(**cfg, not_a_key=1, engine_factory=OracleEngine)
    with pytest.raises(AssertionError, match="Invalid autoencoder variant"):
        train(**dict(cfg, autoencoder_variant="vae"), engine_factory=OracleEngine)
    with pytest.raises(ValueError, match="Invalid optimizer"):
        train(**dict(cfg, optimizer="sgd"), engine_factory=OracleEngine)
    with pytest.raises(ValueError, match="Invalid scheduler"):
        train(**dict(cfg, scheduler="step"), engine_factory=OracleEngine)
    with pytest.raises(KeyError):
        train(**dict(cfg, scheduler="linear", scheduler_params={}), engine_factory=OracleEngine)
    with pytest.raises(NotImplementedError):
        train(**dict(cfg, from_disk=False), engine_factory=OracleEngine)


def test_validation_and_bestval(tmp_path, golden_dir):
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_l1")
    cfg = dict(cfg, val_every=3)
    st = train(**cfg, engine_factory=OracleEngine)
    assert os.path.exists(os.path.join(cfg["run_dir"], "checkpoints", "bestval.pth"))
    assert os.path.exists(os.path.join(cfg["run_dir"], "mo.bestval"))      # the reference's quirk path
    tags = {s["tag"] for s in _scalars(cfg["run_dir"])}
    assert {"val/loss_recon", "val/loss_l1", "val/mse", "val/encoded/num_dead", "val/encoded/percent_dead"} <= tags
    assert np.isfinite(st["best_val_loss"])


def test_topk_train_loop_matches_reference(tmp_path, golden_dir):
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_topk")
    train(**cfg, engine_factory=OracleEngine)
    got = {(s["tag"], s["step"]): s["value"] for s in _scalars(cfg["run_dir"])}
    for tag, val, step in meta["scalars"]:
        assert got[(tag, step)] == pytest.approx(val, rel=2e-5, abs=1e-7), (tag, step)
    ck = torch.load(os.path.join(cfg["run_dir"], "checkpoints", "step7.pth"), map_location="cpu", weights_only=True)
    assert list(ck["model"].keys()) == meta["model_keys"]
    for k in meta["model_keys"]:
        torch.testing.assert_close(ck["model"][k], torch.tensor(z[f"model__{k}"]), rtol=0, atol=2e-7)
    # optimizer state is indexed in model.parameters() order: W_dec, b_dec, encoder.weight, encoder.bias
    order = ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
    for pid, key in enumerate(order):
        st = ck["optimizer"]["state"][pid]
        assert tuple(st["exp_avg"].shape) == tuple(z[f"model__{key}"].shape) == tuple(z[f"opt__{pid}__exp_avg"].shape), key
        assert float(st["step"]) == float(z[f"opt__{pid}__step"])
        ref1, ref2 = z[f"opt__{pid}__exp_avg"], z[f"opt__{pid}__exp_avg_sq"]
        np.testing.assert_allclose(st["exp_avg"].numpy(), ref1, rtol=1e-5, atol=1e-6 * np.abs(ref1).max())
        np.testing.assert_allclose(st["exp_avg_sq"].numpy(), ref2, rtol=1e-5, atol=1e-6 * np.abs(ref2).max())


def test_reference_topk_checkpoint_loads_by_parameter_order(tmp_path, golden_dir):
    """A checkpoint with the reference's own layout (state[0] = W_dec [n,d], state[1] = b_dec [d], state[2] =
    encoder.weight, state[3] = encoder.bias; rebuilt from the arrays the reference's train() saved) resumes into the
    engine with every moment on the right parameter; a mis-ordered one is refused, not copied."""
    from freud_amd.train_sae import load_checkpoint
    z, meta, cfg = _setup(tmp_path, golden_dir, "trainloop_topk")
    d, n = meta["d"], z["model__W_dec"].shape[0]
    order = ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
    ck = {"model": {k: torch.tensor(z[f"model__{k}"]) for k in meta["model_keys"]},
          "optimizer": {"state": {pid: {"step": torch.tensor(float(z[f"opt__{pid}__step"])),
                                        "exp_avg": torch.tensor(z[f"opt__{pid}__exp_avg"]),
                                        "exp_avg_sq": torch.tensor(z[f"opt__{pid}__exp_avg_sq"])} for pid in range(4)},
                        "param_groups": meta["opt_param_groups"]},
          "scheduler": {}, "step": meta["step"], "best_val_loss": meta["best_val_loss"], "hparams": meta["hparams"]}
    path = os.path.join(str(tmp_path), "ref_step7.pth")
    torch.save(ck, path)
    eng = OracleEngine("topk", d, n, 64, optimizer="adam", k=4)
    state = {"engine": eng, "param_order": order}
    load_checkpoint(state, path)
    step, m1, m2 = eng.get_opt_state()
    assert step == 7 and state["step"] == 7
    for pid, key in enumerate(order):
        assert np.array_equal(m1[key], z[f"opt__{pid}__exp_avg"]), key
        assert np.array_equal(m2[key], z[f"opt__{pid}__exp_avg_sq"]), key
        assert np.array_equal(eng.get_params()[key], z[f"model__{key}"]), key
    bad = {"engine": OracleEngine("topk", d, n, 64, optimizer="adam", k=4),
           "param_order": ["encoder.weight", "encoder.bias", "W_dec", "b_dec"]}
    with pytest.raises(ValueError, match="optimizer state"):
        load_checkpoint(bad, path)

