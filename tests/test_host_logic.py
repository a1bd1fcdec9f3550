"""Host-side plugins (monitor / classifier / nudger / dead-weight tracker / config / schedule) against
the golden vectors produced by the REFERENCE's modules.  Runs on CPU: the plugins are exercised on the
oracle's plain-PyTorch module tree (hook fallback path); the fused GPU path is covered in test_host_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

import vae_oracle as vo
from classification.classifier import RegionClassifier
from intervention.nudger import InterventionHandler
from tracking.deadneuron import DeadNeuronTracker
from tracking.monitor import ActivityMonitor

G = os.path.join(os.path.dirname(__file__), "golden")
TARGET = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear, torch.nn.GroupNorm)


@pytest.fixture(scope="module")
def trained():
    """re-runs the golden scenario (4 steps + a validation forward before step 3's monitor.step) with OUR monitor."""
    g = json.load(open(os.path.join(G, "e2e_r32.json")))
    tcfg = json.load(open(os.path.join(G, "tracker.json")))["config"]
    o = vo.OracleWrapper(seed=42)
    tr = vo.OracleTrainer(o, lr=g["lr"], warmup=g["warmup"], max_steps=g["max_steps"], kl_weight=g["kl_weight"])
    mon = ActivityMonitor(o, tcfg)
    logs = {}
    for s in range(1, 5):
        tr.step(vo.synthetic_pixels(g["B"], g["R"], 42, s), vo.synthetic_eps(g["B"], g["R"], 42, s))
        if s == 3:
            o.eval()
            with torch.no_grad():
                o(vo.synthetic_pixels(g["B"], g["R"], 42, 100), sample_posterior=False)
            o.train()
        lg = mon.step(s)
        if lg:
            logs[str(s)] = lg
    return o, mon, logs


def test_monitor_matches_reference_outputs(trained):
    o, mon, logs = trained
    ref = json.load(open(os.path.join(G, "tracker.json")))
    arr = np.load(os.path.join(G, "arrays.npz"))
    assert set(logs) == set(ref["step_logs"]) == {"2", "4"}
    for s, d in ref["step_logs"].items():
        assert set(d) == set(logs[s])  # identical wandb keys: tracking/<layer>.<point>/<metric>_overall_{mean,std}
        for k, v in d.items():
            assert abs(float(logs[s][k]) - v) <= 2e-5 * abs(v) + 1e-9, (s, k)
    for key in arr.files:
        if not key.startswith("track/"):
            continue
        _, s, rest = key.split("/", 2)
        lid, metric = rest.rsplit("/", 1)
        got = np.asarray(mon.get_data_for_step(int(s))[lid][metric])
        np.testing.assert_allclose(got, arr[key], rtol=2e-5, atol=1e-7)
    recs = mon.export_all_processed_data_to_records()
    assert [(r["global_step"], r["layer_identifier"], r["original_metric_name"], r["metric_type"]) for r in recs] == \
           [(r["global_step"], r["layer_identifier"], r["original_metric_name"], r["metric_type"]) for r in ref["records"]]
    for a, b in zip(recs, ref["records"]):
        assert abs(a["metric_value"] - b["metric_value"]) <= 2e-5 * abs(b["metric_value"]) + 1e-9


def test_classifier_mask_is_bit_identical_on_golden_vectors(trained):
    o, _, _ = trained
    ref = json.load(open(os.path.join(G, "classifier.json")))
    arr = np.load(os.path.join(G, "arrays.npz"))
    data4 = {}
    for key in arr.files:
        if key.startswith("track/4/"):
            lid, metric = key[len("track/4/"):].rsplit("/", 1)
            data4.setdefault(lid, {})[metric] = arr[key]
    for case, d in ref.items():
        lid = case.split("@")[0]
        cfg = {"enabled": True, "method": "threshold_groupnorm_activity", "threshold": d["threshold"],
               "target_metric_key": "mean_abs_activation_per_channel", "layers_to_classify": [lid]}
        got = RegionClassifier(o.vae, cfg).classify(data4, 4)
        assert got == d["result"], case  # indices, values, param names, threshold: exact
    # disabled / unknown method / empty data (classifier.py:101-112)
    assert RegionClassifier(o.vae, {"enabled": False}).classify(data4, 4) == {}
    assert RegionClassifier(o.vae, {"enabled": True, "method": "other"}).classify(data4, 4) == {}
    assert RegionClassifier(o.vae, {"enabled": True}).classify({}, 4) == {}
    # GN map: plain + 'vae.' alias, conv layers are never classified (classifier.py:43-95)
    c = RegionClassifier(o.vae, {"enabled": True, "threshold": 1e9})
    assert c._lookup_param_info("vae.encoder.conv_norm_out.output") == ("encoder.conv_norm_out.weight", 512)
    assert c._lookup_param_info("encoder.conv_in.output") is None
    assert len(c._layer_to_param_map) == 2 * 52


def test_nudger_is_bit_identical(trained):
    o, _, _ = trained
    ref = json.load(open(os.path.join(G, "nudger.json")))
    arr = np.load(os.path.join(G, "arrays.npz"))
    pname = "encoder.down_blocks.0.resnets.0.norm1.weight"
    p = o.vae.get_parameter(pname)
    keep = p.detach().clone()
    for key, d in ref.items():
        strat, factor, cap = key.split("/")
        with torch.no_grad():
            p.copy_(torch.from_numpy(arr["nudger/gamma0"]))
        h = InterventionHandler(o.vae, {"enabled": True, "strategy": strat, "nudge_factor": float(factor),
                                        "max_scale_value": float(cap), "intervention_interval": 20})
        res = {"layer": {"param_name_scale": pname, "inactive_channel_indices": d["indices"]}}
        h.intervene(res, 10)
        assert np.array_equal(p.detach().numpy(), arr["nudger/gamma0"])  # gate: step % interval != 0
        h.intervene(res, 0)
        assert np.array_equal(p.detach().numpy(), arr["nudger/gamma0"])  # gate: step 0
        h.intervene(res, 20)
        h.intervene(res, 40)
        assert np.array_equal(p.detach().numpy(), arr[f"nudger/{key}"]), key  # bit-exact incl. cap and fp64 product rounding
        assert h.num_nudges_applied == d["num_nudges_applied"]
    # the reference's own KAT (nudger.py:256-258): 1.0 -> min(1.0*1.2, 1.5) = 1.2 on idx [0,2,5,15]
    net = torch.nn.Sequential()
    net.gn = torch.nn.GroupNorm(4, 16)
    h = InterventionHandler(net, {"enabled": True, "strategy": "gentle_nudge_groupnorm_scale", "nudge_factor": 1.2,
                                  "max_scale_value": 1.5, "intervention_interval": 1})
    h.intervene({"x": {"param_name_scale": "gn.weight", "inactive_channel_indices": [0, 2, 5, 15, 99]}}, 1)
    w = net.gn.weight.detach()
    assert torch.equal(w[[0, 2, 5, 15]], torch.full((4,), 1.2)) and float(w[1]) == 1.0 and h.num_nudges_applied == 4
    # duplicates compound sequentially exactly as the reference loop does
    h.intervene({"x": {"param_name_scale": "gn.weight", "inactive_channel_indices": [1, 1]}}, 2)
    assert float(net.gn.weight[1].detach()) == float(torch.tensor(min(float(torch.tensor(1.2, dtype=torch.float32)) * 1.2, 1.5), dtype=torch.float32))
    with torch.no_grad():
        p.copy_(keep)


def test_dead_weight_percentages_match_reference():
    ref = json.load(open(os.path.join(G, "deadneuron.json")))
    arr = np.load(os.path.join(G, "arrays.npz"))
    o = vo.OracleWrapper(seed=42)  # fresh synthetic weights + planted values, as in make_golden.py
    with torch.no_grad():
        w = o.vae.get_parameter("decoder.conv_out.weight")
        w.view(-1)[:100] = 0.0
        w.view(-1)[100:200] = 5e-6
        o.vae.get_parameter("encoder.mid_block.attentions.0.to_k.bias").zero_()
    for mode in ("threshold", "percent_of_mean", "both"):
        t = DeadNeuronTracker(TARGET, ["encoder.conv_in.weight"], threshold=1e-5, mean_percentage=0.1, dead_type=mode)
        t.track_dead_neurons(o, 7)  # wrapper with .vae (deadneuron.py:38-44)
        got = {k: v[0][1] for k, v in t.percent_history.items()}
        assert set(got) == set(ref[mode])
        for k, v in ref[mode].items():
            assert got[k] == pytest.approx(v, rel=1e-12, abs=0), (mode, k)
        assert all(v[0][0] == 7 for v in t.percent_history.values())
        if mode == "threshold":
            np.testing.assert_array_equal(t.weights_history["encoder.conv_in.weight"][0], arr["dead/raw/encoder.conv_in.weight"])
    # the reference's own KAT, same dummy values and expected numbers (deadneuron.py:122-202)
    class DummyVAE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 8, kernel_size=3, padding=1)  # 216 elements
            self.conv1.weight.data.fill_(0.001)
            self.conv1.weight.data[0, 0, 0, 0] = 1.0
            self.conv1.weight.data[1, 0, 0, 0] = 1e-7
            self.gn1 = torch.nn.GroupNorm(2, 8)
            self.gn1.weight.data.fill_(1e-6)
            self.gn1.bias.data.fill_(1e-7)

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.vae = DummyVAE()

    m = Wrapper()
    t = DeadNeuronTracker((torch.nn.Conv2d, torch.nn.Linear, torch.nn.GroupNorm), ["gn1.weight"], 1e-5, 0.1, "both")
    t.track_dead_neurons(m, 0)
    m.vae.conv1.weight.data.fill_(1.0)
    m.vae.gn1.weight.data.fill_(1.0)
    m.vae.gn1.bias.data.fill_(0.5)
    t.track_dead_neurons(m, 20)
    assert t.percent_history["conv1.weight"] == [(0, (1 / 216) * 100.0), (20, 0.0)]
    assert t.percent_history["gn1.weight"] == [(0, 0.0), (20, 0.0)]
    assert t.percent_history["gn1.bias"] == [(0, 0.0), (20, 0.0)]
    assert t.weights_history["gn1.weight"][0].shape == (8,)
    assert DeadNeuronTracker(TARGET, [], 1e-5, 0.1, "bogus").get_percentage(m.vae.conv1.weight) == 0.0


def test_config_loader_shallow_merge(tmp_path):
    from utils.config_utils import load_config
    (tmp_path / "base.yaml").write_text("seed: 1\ntraining:\n  lr_warmup_steps: 7\n  learning_rate: 1.0e-5\nrun_name: base\n")
    (tmp_path / "exp.yaml").write_text("defaults: [base]\nrun_name: exp\ntraining:\n  learning_rate: 5.0e-5\n")
    c = load_config(str(tmp_path / "exp.yaml"))
    assert c["seed"] == 1 and c["run_name"] == "exp" and "defaults" not in c
    assert c["training"] == {"learning_rate": 5e-5}  # nested dict REPLACED: base lr_warmup_steps is dropped (config_utils.py:55)
    with pytest.raises(FileNotFoundError):
        load_config(str(tmp_path / "nope.yaml"))
    (tmp_path / "bad.yaml").write_text("defaults: [missing]\n")
    with pytest.raises(FileNotFoundError):
        load_config(str(tmp_path / "bad.yaml"))


def test_lr_schedule_and_data_pipeline(tmp_path):
    from vaehip.trainer import lr_lambda_factory
    from data_utils import load_and_preprocess_dataset, create_dataloader, get_transform, safe_collate
    f = lr_lambda_factory(100, 1000)
    assert [f(s) for s in (0, 50, 100, 550, 1000, 2000)] == [vo.lr_lambda(s, 100, 1000) for s in (0, 50, 100, 550, 1000, 2000)]
    ds = load_and_preprocess_dataset("synthetic:10", resolution=16, max_samples=6)
    assert len(ds) == 6
    b = next(iter(create_dataloader(ds, 4, shuffle=False)))
    assert b["pixel_values"].shape == (4, 3, 16, 16) and float(b["pixel_values"].abs().max()) <= 1.0
    assert torch.equal(ds[3]["pixel_values"], ds[3]["pixel_values"])
    assert safe_collate([{"pixel_values": None}]) is None
    from PIL import Image
    d = tmp_path / "imgs" / "train"
    d.mkdir(parents=True)
    Image.fromarray((np.arange(40 * 30 * 3) % 255).astype(np.uint8).reshape(30, 40, 3)).save(d / "a.png")
    Image.fromarray(np.full((20, 20), 128, np.uint8)).save(d / "b.png")  # grayscale -> RGB
    ds2 = load_and_preprocess_dataset(str(tmp_path / "imgs"), resolution=16, split="train")
    assert len(ds2) == 2 and ds2[0]["pixel_values"].shape == (3, 16, 16)
    g = ds2[1]["pixel_values"]
    assert torch.allclose(g, torch.full_like(g, 128 / 255 * 2 - 1), atol=1e-6)
    assert get_transform(8)(Image.new("RGB", (8, 8), (255, 0, 0)))[0].min() == 1.0


def test_accumulation_window_is_flushed_at_the_end_of_a_dataloader_pass():
    """accelerate.accumulate forces an optimizer update on the last batch of every dataloader pass (accelerator.py
    _do_sync), so ceil(len/accum) updates happen per epoch -- the count train.py:188-195 schedules.  Host logic only:
    the engine, the add kernel and the optimizer kernels are replaced by stand-ins (they need a GPU)."""
    import math
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    from train import with_last
    from vaehip.trainer import HipTrainer
    assert list(with_last([])) == [] and list(with_last("abc")) == [("a", False), ("b", False), ("c", True)]
    w = SDXLVAEWrapper("synthetic:1")
    n_batches, accum, epochs = 5, 2, 3
    upd_per_epoch = math.ceil(n_batches / accum)
    tr = HipTrainer(w, lr=1.0, lr_warmup_steps=0, max_train_steps=epochs * upd_per_epoch, gradient_accumulation_steps=accum)
    grad = w.vae.arena.grad
    seen = []  # gradient sums the optimizer saw

    def fake_fwd_bwd(pv, eps, klw, sample, gen, grad_scale=1.0):
        grad.fill_(float(pv) * grad_scale)
        return {"scalars": torch.zeros(3)}
    w.vae.engine.forward_backward = fake_fwd_bwd
    tr._add = lambda a, b, out: torch.add(a, b, out=out)
    tr.optimizer.step = lambda: seen.append(float(grad[0]))
    updates = 0
    for _ in range(epochs):
        for b, last in with_last(range(1, n_batches + 1)):
            tr.train_step(b, None, end_of_dataloader=last)
            updates += tr.sync_gradients
        assert tr.pending_micro_batches == 0  # nothing carries into the next epoch
    assert updates == tr.global_step == epochs * upd_per_epoch
    assert tr.lr_scheduler.get_last_lr()[0] == 0.0  # the LR reached the end of its linear decay
    # windows (1,2) (3,4) (5): every micro-batch scaled by 1/accum, the partial window too
    assert seen == [1.5, 3.5, 2.5] * epochs
    tr.train_step(7, None)
    assert tr.pending_micro_batches == 1
    tr.flush()
    assert tr.pending_micro_batches == 0 and seen[-1] == 3.5 and tr.global_step == epochs * upd_per_epoch + 1


def test_fp32_activation_image_policy(monkeypatch):
    """which 3x3 layers read a materialised GroupNorm+SiLU tensor in fp32 mode (ops.act_image32_ok): the shapes the Winograd
    kernels take, from ACT_IMAGE32_MIN_CIN input channels on; never in bf16 mode, never with the direct kernels selected"""
    from vaehip import ops
    assert ops.PRECISION == ops.PREC_F32 and ops.WINOGRAD
    assert ops.get_option("no_wino") == 0 and ops.get_option("flat_conv") == 0 and ops.get_option("bogus") == -1
    # "wide_reserved_cus" is a count (CUs the persistent bf16 kernel leaves to RCCL), range-checked by the library
    assert ops.get_option("no_wgrad_dma") == 0 and ops.get_option("no_thin_mfma") == 0 and ops.get_option("no_wino4") == 0
    with ops.option("no_wgrad_dma"):
        assert ops.get_option("no_wgrad_dma") == 1
    assert ops.get_option("wide_reserved_cus") == 0
    with ops.option("wide_reserved_cus", 32):
        assert ops.get_option("wide_reserved_cus") == 32
    assert ops.get_option("wide_reserved_cus") == 0
    with pytest.raises(RuntimeError, match="wide_reserved_cus"):
        ops.lib.call("vae_set_option", b"wide_reserved_cus", 500)
    ok = lambda kind, shape, co, ci: ops.act_image32_ok(kind, shape, co, ci)
    assert ok("c3", (2, 32, 32, 128), 128, 128) and ok("c3", (16, 64, 64, 512), 512, 512) and ok("c3", (1, 8, 16, 256), 64, 256)
    assert not ok("c3", (2, 32, 32, 64), 128, 64)        # below the channel threshold
    assert not ok("c1", (2, 32, 32, 128), 128, 128)      # 1x1 / stride-2 / upsampler convolutions keep the fused transform
    assert not ok("c3", (2, 12, 32, 128), 128, 128) and not ok("c3", (2, 32, 24, 128), 128, 128)  # H % 8, W % 16
    assert not ok("c3", (2, 32, 32, 128), 3, 128)        # conv_out: 3 output channels
    monkeypatch.setattr(ops, "ACT_IMAGE32_MIN_CIN", 256)
    assert not ok("c3", (2, 32, 32, 128), 128, 128) and ok("c3", (2, 32, 32, 256), 256, 256)
    monkeypatch.setattr(ops, "ACT_IMAGE32_MIN_CIN", 128)
    monkeypatch.setattr(ops, "WINOGRAD", False)
    assert not ok("c3", (2, 32, 32, 128), 128, 128)
    monkeypatch.setattr(ops, "WINOGRAD", True)
    with ops.option("no_wino"):  # the library-side switch (tests compare the two fp32 algorithms with it)
        assert ops.get_option("no_wino") == 1 and not ok("c3", (2, 32, 32, 128), 128, 128)
    with ops.option("flat_conv"):
        assert not ok("c3", (2, 32, 32, 128), 128, 128)
    assert ops.get_option("no_wino") == 0 and ok("c3", (2, 32, 32, 128), 128, 128)
    with ops.precision(ops.PREC_BF16):
        assert not ok("c3", (2, 32, 32, 128), 128, 128)
