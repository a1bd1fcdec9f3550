"""GPU product path against the golden vectors the REFERENCE's modules produced (tests/golden):
the fused train step + fused tracker + classifier + nudger + dead-weight scan, through the C ABI."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear, torch.nn.GroupNorm)


@pytest.fixture(scope="module")
def scenario(cuda):
    """the golden scenario: 4 train steps, a validation forward before step 3's monitor.step()."""
    import vae_oracle as vo
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    from tracking.monitor import ActivityMonitor
    from vaehip.trainer import HipTrainer
    g = json.load(open(os.path.join(G, "e2e_r32.json")))
    tcfg = json.load(open(os.path.join(G, "tracker.json")))["config"]
    w = SDXLVAEWrapper("synthetic:1")
    w.vae.load_state_dict(vo.synthetic_state_dict(vo.OracleAutoencoderKL(), 42))
    w.to(cuda)
    tr = HipTrainer(w, lr=g["lr"], lr_warmup_steps=g["warmup"], max_train_steps=g["max_steps"], kl_weight=g["kl_weight"],
                    max_grad_norm=1.0)
    mon = ActivityMonitor(w, tcfg)
    assert len(mon.fused_layers) == 3  # the 3 shipped layers are fused; the multi-metric one uses the hook slow path
    steps, logs, val = [], {}, None
    for s in range(1, 5):
        lr = tr.optimizer.param_groups[0]["lr"]
        res = tr.train_step(vo.synthetic_pixels(g["B"], g["R"], 42, s).to(cuda), vo.synthetic_eps(g["B"], g["R"], 42, s).to(cuda))
        sc = res["scalars"].cpu().tolist()
        steps.append({"rec": sc[0], "kl": sc[1], "total": sc[2], "grad_norm": tr.optimizer.grad_norm().item(), "lr": lr})
        if s == 3:
            w.eval()
            r = tr.eval_step(vo.synthetic_pixels(g["B"], g["R"], 42, 100).to(cuda))
            val = {"rec_sum": r["rec_sum"].item(), "kl_sum": r["kl_sum"].item()}
            w.train()
        lg = mon.step(s)
        if lg:
            logs[str(s)] = lg
    return g, w, tr, mon, steps, logs, val


def _record(name, data):
    """measured deviations -> gpurun_out/host_gpu_measured.json (copied to profiles/ by hand when it is to be judged)"""
    path = os.path.join(ROOT, "gpurun_out", "host_gpu_measured.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name] = data
        json.dump(cur, open(path, "w"), indent=1)
    except OSError:
        pass


# Tolerances after optimizer updates are CALIBRATED, not asserted (tools/calibrate_post_update_tolerances.py ->
# profiles/r04_post_update_drift.json): the fp32 CPU oracle drifts from the same scenario in float64 by <= 3.8e-5 on every scalar
# of steps 1-4 (the gradient norm; the KL by 1.2e-5 at step 3, 1e-7 before any update), 7.5e-6 on the validation sums and
# 1.2e-5 on the step-4 tracker vectors, so two correct fp32 implementations differ by <= 2 x that < 1e-4: north_star's 1e-4
# holds for all four steps and no extra headroom is granted (round 3 allowed 1e-3 / 2e-3 without a measurement).
TOL_SCALARS = 1e-4
TOL_VALIDATION = 1e-4
TOL_TRACKER = 1e-4


def test_train_steps_match_golden(scenario):
    g, w, tr, mon, steps, logs, val = scenario
    meas = []
    for s, (got, ref) in enumerate(zip(steps, g["steps"]), start=1):
        meas.append({k: abs(got[k] - ref[k]) / abs(ref[k]) for k in ("rec", "kl", "total", "grad_norm")})
    vm = {k: abs(val[k] - g["val"][k]) / abs(g["val"][k]) for k in ("rec_sum", "kl_sum")}
    chk = float(w.vae.arena.flat.double().abs().sum())
    _record("train_steps_vs_golden_rel", {"steps": meas, "validation": vm,
                                         "param_abs_checksum": abs(chk - g["param_abs_checksum_after"]) / g["param_abs_checksum_after"]})
    for s, (got, ref) in enumerate(zip(steps, g["steps"]), start=1):
        for k in ("rec", "kl", "total", "grad_norm"):
            assert abs(got[k] - ref[k]) <= TOL_SCALARS * abs(ref[k]), (s, k, got[k], ref[k])
        assert got["lr"] == pytest.approx(ref["lr"], rel=1e-12, abs=1e-15)
    assert val["rec_sum"] == pytest.approx(g["val"]["rec_sum"], rel=TOL_VALIDATION)
    assert val["kl_sum"] == pytest.approx(g["val"]["kl_sum"], rel=TOL_VALIDATION)
    assert chk == pytest.approx(g["param_abs_checksum_after"], rel=1e-5)


def test_fused_tracker_and_classifier_match_reference(scenario):
    from classification.classifier import RegionClassifier
    g, w, tr, mon, steps, logs, val = scenario
    ref = json.load(open(os.path.join(G, "tracker.json")))
    arr = np.load(os.path.join(G, "arrays.npz"))
    assert set(logs) == {"2", "4"}
    for s, d in ref["step_logs"].items():
        assert set(d) == set(logs[s])
    worst, per_key = 0.0, {}
    for key in arr.files:
        if not key.startswith("track/"):
            continue
        _, s, rest = key.split("/", 2)
        lid, metric = rest.rsplit("/", 1)
        got = np.asarray(mon.get_data_for_step(int(s))[lid][metric], dtype=np.float64)
        refv = arr[key].astype(np.float64)
        rel = float(np.max(np.abs(got - refv) / (np.abs(refv) + 1e-12)))
        worst = max(worst, rel)
        per_key[key] = rel
    _record("tracker_vs_reference_rel", per_key)
    for key, rel in per_key.items():
        assert rel < TOL_TRACKER, (key, rel)
    print("worst tracker rel err", worst)
    # step-2 statistics (before any parameter divergence): inactivity masks identical to the reference classifier's
    data2 = mon.get_data_for_step(2)
    for lid in ["vae.encoder.down_blocks.0.resnets.0.norm1.output", "vae.decoder.up_blocks.1.resnets.0.norm1.output"]:
        refv = arr[f"track/2/{lid}/mean_abs_activation_per_channel"]
        for q in (0.1, 0.5, 0.9):
            thr = float(np.quantile(refv, q))
            gap = np.min(np.abs(refv - np.float32(thr)))
            if gap < 1e-4 * abs(thr):
                thr = float(np.float32(thr) + np.float32(2e-4 * abs(thr)))  # keep the threshold off a data point
            cfg = {"enabled": True, "threshold": thr, "layers_to_classify": [lid]}
            c = RegionClassifier(w.vae, cfg)
            a = c.classify(data2, 2)
            b = c.classify({lid: {"mean_abs_activation_per_channel": refv}}, 2)
            assert a[lid]["inactive_channel_indices"] == b[lid]["inactive_channel_indices"], (lid, q)
            assert a[lid]["param_name_scale"] == lid[len("vae."):-len(".output")] + ".weight"
    recs = mon.export_all_processed_data_to_records()
    assert [(r["global_step"], r["layer_identifier"], r["metric_type"]) for r in recs] == \
           [(r["global_step"], r["layer_identifier"], r["metric_type"]) for r in ref["records"]]


def test_classification_at_thresholds_on_the_data_measured(scenario):
    """north_star asks for a bit-identical inactivity mask.  The test above keeps thresholds >= 2e-4 away from any statistic;
    this one does NOT: thresholds at plain quantiles of the reference statistics (no offset) and thresholds EQUAL to a data
    point (classifier.py:135 is a strict `<` in fp32).  It records every channel whose GPU classification differs from the
    reference classifier's on the reference's own statistics (gpurun_out/classifier_flips.json, committed under profiles/),
    and requires of a flipped channel only what arithmetic can give: its two statistics agree to a few fp32 ulps and the
    threshold lies between them (a flip anywhere else is a real error)."""
    from classification.classifier import RegionClassifier
    g, w, tr, mon, steps, logs, val = scenario
    arr = np.load(os.path.join(G, "arrays.npz"))
    data2 = mon.get_data_for_step(2)
    record = {"layers": {}, "thresholds": 0, "channels_compared": 0, "flips": []}
    for lid in ["vae.encoder.conv_in.output", "vae.encoder.down_blocks.0.resnets.0.norm1.output",
                "vae.decoder.up_blocks.1.resnets.0.norm1.output"]:
        refv = arr[f"track/2/{lid}/mean_abs_activation_per_channel"]
        gotv = np.asarray(data2[lid]["mean_abs_activation_per_channel"], dtype=np.float32)
        ulps = np.abs(gotv.view(np.int32).astype(np.int64) - refv.view(np.int32).astype(np.int64))
        record["layers"][lid] = {"channels": int(refv.size), "bitwise_equal_channels": int((ulps == 0).sum()), "max_ulp_distance": int(ulps.max()),
                                 "max_rel_err": float(np.max(np.abs(gotv.astype(np.float64) - refv) / np.abs(refv)))}
        srt = np.sort(refv)
        thrs = [float(np.quantile(refv, q)) for q in (0.1, 0.25, 0.5, 0.75, 0.9)]          # interpolated quantiles, no offset
        thrs += [float(srt[int(q * (srt.size - 1))]) for q in (0.1, 0.25, 0.5, 0.75, 0.9)]  # exactly ON a data point
        for thr in thrs:
            t32 = np.float32(thr)
            a = gotv < t32   # classifier.py:135 under NumPy 2: fp32 compare against the fp32-rounded threshold
            b = refv < t32
            record["thresholds"] += 1
            record["channels_compared"] += int(refv.size)
            if lid != "vae.encoder.conv_in.output":  # conv outputs have no GroupNorm scale to map to; the classifier proper on the rest
                c = RegionClassifier(w.vae, {"enabled": True, "threshold": thr, "layers_to_classify": [lid]})
                assert c.classify(data2, 2)[lid]["inactive_channel_indices"] == np.where(a)[0].tolist()
            for ch in np.where(a != b)[0]:
                record["flips"].append({"layer": lid, "threshold": thr, "channel": int(ch), "gpu": float(gotv[ch]), "reference": float(refv[ch]),
                                        "ulp_distance": int(ulps[ch])})
                lo, hi = min(gotv[ch], refv[ch]), max(gotv[ch], refv[ch])
                assert lo <= t32 <= hi and ulps[ch] <= 64, (lid, thr, ch, gotv[ch], refv[ch])
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(record, open(os.path.join(out, "classifier_flips.json"), "w"), indent=1)
    print("classification on-threshold record:", json.dumps({k: v for k, v in record.items() if k != "flips"}), "flips:", len(record["flips"]))


def test_nudger_on_live_arena_is_bit_identical(scenario, cuda):
    from intervention.nudger import InterventionHandler
    g, w, tr, mon, *_ = scenario
    ref = json.load(open(os.path.join(G, "nudger.json")))
    arr = np.load(os.path.join(G, "arrays.npz"))
    pname = "encoder.down_blocks.0.resnets.0.norm1.weight"
    p = w.vae.get_parameter(pname)
    keep = p.detach().clone()
    for key, d in ref.items():
        strat, factor, cap = key.split("/")
        with torch.no_grad():
            p.copy_(torch.from_numpy(arr["nudger/gamma0"]).to(cuda))
        h = InterventionHandler(w.vae, {"enabled": True, "strategy": strat, "nudge_factor": float(factor),
                                        "max_scale_value": float(cap), "intervention_interval": 20})
        res = {"layer": {"param_name_scale": pname, "inactive_channel_indices": d["indices"]}}
        h.intervene(res, 20)
        h.intervene(res, 40)
        assert np.array_equal(p.detach().cpu().numpy(), arr[f"nudger/{key}"]), key
        assert w.vae.arena.owns(w.vae)  # edited in place: still the arena the kernels read
    with torch.no_grad():
        p.copy_(keep)


def test_dead_weight_scan_kernel_matches_reference(cuda):
    import vae_oracle as vo
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    from tracking.deadneuron import DeadNeuronTracker
    ref = json.load(open(os.path.join(G, "deadneuron.json")))
    w = SDXLVAEWrapper("synthetic:1")
    w.vae.load_state_dict(vo.synthetic_state_dict(vo.OracleAutoencoderKL(), 42))  # fresh weights, as the fixture
    w.to(cuda)
    cw = w.vae.get_parameter("decoder.conv_out.weight")
    with torch.no_grad():  # plant in LOGICAL (O,I,H,W) order like the fixture did
        flat = cw.detach().cpu().contiguous().view(-1)
        flat[:100] = 0.0
        flat[100:200] = 5e-6
        cw.copy_(flat.view(cw.shape).to(cuda))
        w.vae.get_parameter("encoder.mid_block.attentions.0.to_k.bias").zero_()
    for mode in ("threshold", "percent_of_mean", "both"):
        t = DeadNeuronTracker(TARGET, ["encoder.conv_in.weight"], threshold=1e-5, mean_percentage=0.1, dead_type=mode)
        t.track_dead_neurons(w, 7)
        got = {k: v[0][1] for k, v in t.percent_history.items()}
        assert set(got) == set(ref[mode])
        for k, v in ref[mode].items():
            assert got[k] == pytest.approx(v, rel=1e-9, abs=1e-12), (mode, k, got[k], v)


def test_train_and_evaluate_cli_plumbing(cuda, tmp_path):
    """the drop-in entry points end to end on a synthetic dataset (no network): files the reference writes exist."""
    src = os.path.join(ROOT, "vae-channel-dynamics_amd", "src")
    cfg = os.path.join(ROOT, "vae-channel-dynamics_amd", "configs", "experiment_synthetic_test.yaml")
    import yaml
    c = yaml.safe_load(open(cfg))
    c["output_dir"] = str(tmp_path)
    c["data"]["dataset_name"] = "synthetic:48"  # 6 steps/epoch x 2 epochs: tracker fires at step 10
    c["data"]["resolution"] = 32
    cpath = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(c, open(cpath, "w"))
    env = dict(os.environ, PYTHONPATH=src)
    r = subprocess.run([sys.executable, os.path.join(src, "train.py"), "--config_path", cpath], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    run = tmp_path / c["run_name"]
    for f in ["config.yaml", "metrics.jsonl", "tracked_activation_stats.csv", "dead_neuron_percentage_history.csv",
              "final_model/model.safetensors", "final_model/optimizer.bin", "final_model/scheduler.bin",
              "final_model/random_states_0.pkl", "final_model/vae/config.json", "final_model/vae/diffusion_pytorch_model.safetensors"]:
        assert (run / f).exists(), f
    lines = [json.loads(l) for l in open(run / "metrics.jsonl")]
    assert any("train_loss_step" in l for l in lines) and any("validation/avg_total_loss" in l for l in lines)
    assert any(k.startswith("tracking/vae.encoder.conv_in.output/") for l in lines for k in l)
    r = subprocess.run([sys.executable, os.path.join(src, "evaluate.py"), "--config_path", cpath, "--checkpoint_path",
                        str(run / "final_model"), "--eval_split", "validation", "--num_samples_to_save", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    txt = open(run / "final_model" / "eval_results_validation" / "eval_metrics.txt").read()
    assert "Average MSE:" in txt and "Average KL:" in txt and "Average SSIM:" in txt
    assert (run / "final_model" / "eval_results_validation" / "sample_1_recon.png").exists()

    # the NUMBERS evaluate.py prints (reference src/evaluate.py:216-240,314-326) against the CPU oracle's
    # deterministic forward on the same checkpoint and the same split, aggregated the same way (per-batch means weighted by
    # the batch size; PSNR from the summed squared error of the [0,1]-clamped images; SSIM = mean of per-image indices,
    # through evaluate.py's own float64 window formulas).  Parity unpinned vs torchmetrics (absent; the reference holds no
    # fixture): what is pinned here is GPU forward == oracle forward through the whole metrics path, to 1e-4 relative.
    import importlib
    import vae_oracle as vo
    from safetensors.torch import load_file
    sys.path.insert(0, src)
    ev = importlib.import_module("evaluate")
    from data_utils import load_and_preprocess_dataset
    got = {}
    for line in txt.splitlines():
        k, _, v = line.partition(": ")
        if k.startswith("Average") or k.startswith("Number"):
            got[k] = float(v)
    o = vo.OracleAutoencoderKL()
    o.load_state_dict(load_file(str(run / "final_model" / "vae" / "diffusion_pytorch_model.safetensors")))
    o.eval()
    ds = load_and_preprocess_dataset(dataset_name=c["data"]["dataset_name"], dataset_config_name=None, image_column="image",
                                     resolution=32, max_samples=c["data"]["validation_max_samples"], split="validation")
    bs = c["data"]["validation_batch_size"]
    n = 0
    tm = tk = sse = ssim = 0.0
    cnt = 0
    with torch.no_grad():
        for i0 in range(0, len(ds), bs):
            pv = torch.stack([ds[i]["pixel_values"] for i in range(i0, min(len(ds), i0 + bs))])
            d = o.encode(pv).latent_dist
            rec = o.decode(d.mode()).sample
            b = pv.shape[0]
            tm += torch.nn.functional.mse_loss(rec, pv).item() * b
            tk += d.kl().mean().item() * b
            r01, o01 = ev.to_unit(rec), ev.to_unit(pv)
            s_, c_ = ev.psnr_sums(r01, o01)
            sse += float(s_)
            cnt += c_
            ssim += float(ev.ssim_per_image(r01, o01).double().sum())
            n += b
    want = {"Number of Samples Processed": n, "Average MSE": tm / n, "Average KL": tk / n,
            "Average PSNR": 10.0 * math.log10(1.0 / (sse / cnt)), "Average SSIM": ssim / n}
    assert n == 16
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-4), (k, got[k], v)
