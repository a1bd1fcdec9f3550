"""Generates tests/golden/*.json|*.npz by running the REFERENCE's own torch-only plugin modules
(/root/reference/src/{tracking,classification,intervention}) against the CPU oracle's module tree.

Run in the build container only (the reference never travels):   python tests/golden/make_golden.py
The fixtures are data (inputs are regenerated from the oracle's counter-hash generator, outputs are
stored); no reference source is copied.  What they pin:
  e2e_r32        3 train steps (train.py:283-306 semantics) + 1 validation forward: losses, grad norm, lr, tracker
  tracker        ActivityMonitor.step() dict / get_data_for_step vectors / CSV records (monitor.py)
  classifier     RegionClassifier.classify output at thresholds straddling the data (classifier.py)
  nudger         InterventionHandler gamma before/after (nudger.py), incl. cap and fp64-product rounding
  deadneuron     DeadNeuronTracker percentages for the three modes + the reference's own KAT 1/216*100
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference/src")

import vae_oracle as vo  # noqa: E402
from tracking.monitor import ActivityMonitor as RefMonitor  # noqa: E402
from tracking.deadneuron import DeadNeuronTracker as RefDNT  # noqa: E402
from classification.classifier import RegionClassifier as RefClassifier  # noqa: E402
from intervention.nudger import InterventionHandler as RefNudger  # noqa: E402

TRACK_CFG = {
    "enabled": True, "track_interval": 2,
    "target_layers": [
        {"name": "vae.encoder.conv_in", "capture_point": "output", "metrics": ["mean_abs_activation_per_channel"]},
        {"name": "vae.encoder.down_blocks.0.resnets.0.norm1", "capture_point": "output", "metrics": ["mean_abs_activation_per_channel"]},
        {"name": "vae.decoder.up_blocks.1.resnets.0.norm1", "capture_point": "output", "metrics": ["mean_abs_activation_per_channel"]},
        {"name": "vae.decoder.conv_norm_out", "capture_point": "input", "metrics": ["mean_abs_activation_per_channel", "mean_activation", "std_activation"]},
    ],
}
R, B, KLW, STEPS = 32, 2, 1e-6, 4
TARGET_CLASSES = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear, torch.nn.GroupNorm)


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    return o


def main():
    torch.manual_seed(0)
    o = vo.OracleWrapper(seed=42)
    tr = vo.OracleTrainer(o, lr=1e-4, warmup=2, max_steps=10, kl_weight=KLW, max_grad_norm=1.0)
    mon = RefMonitor(o, TRACK_CFG)
    arrays = {}
    e2e = {"R": R, "B": B, "kl_weight": KLW, "lr": 1e-4, "warmup": 2, "max_steps": 10, "steps": []}
    tracker = {"config": TRACK_CFG, "step_logs": {}, "records": None}
    for s in range(1, STEPS + 1):
        x, eps = vo.synthetic_pixels(B, R, 42, s), vo.synthetic_eps(B, R, 42, s)
        rec = tr.step(x, eps)
        e2e["steps"].append(rec)
        if s == 3:  # a validation forward between tracked steps pollutes the buffer (train.py:75, monitor.py:98-101)
            o.eval()
            with torch.no_grad():
                out = o(vo.synthetic_pixels(B, R, 42, 100), sample_posterior=False)
                e2e["val"] = {"rec_sum": float(torch.nn.functional.mse_loss(out["reconstruction"], vo.synthetic_pixels(B, R, 42, 100), reduction="sum")),
                              "kl_sum": float(out["latent_dist"].kl().sum())}
            o.train()
        logs = mon.step(s)
        if logs:
            tracker["step_logs"][str(s)] = jsonable(logs)
            for lid, metrics in mon.get_data_for_step(s).items():
                for mname, val in metrics.items():
                    arrays[f"track/{s}/{lid}/{mname}"] = np.asarray(val)
    tracker["records"] = jsonable(mon.export_all_processed_data_to_records())
    e2e["param_checksum_after"] = float(sum(p.detach().double().sum() for p in o.parameters()))
    e2e["param_abs_checksum_after"] = float(sum(p.detach().double().abs().sum() for p in o.parameters()))

    # ---- classifier at thresholds straddling the step-4 data
    data4 = mon.get_data_for_step(4)
    cls_out = {}
    for lid in ["vae.encoder.down_blocks.0.resnets.0.norm1.output", "vae.decoder.up_blocks.1.resnets.0.norm1.output"]:
        v = data4[lid]["mean_abs_activation_per_channel"]
        for q in (0.1, 0.5):
            thr = float(np.quantile(v, q))
            cfg = {"enabled": True, "method": "threshold_groupnorm_activity", "threshold": thr,
                   "target_metric_key": "mean_abs_activation_per_channel", "layers_to_classify": [lid]}
            cls_out[f"{lid}@{q}"] = {"threshold": thr, "result": jsonable(RefClassifier(o.vae, cfg).classify(data4, 4))}
        # edge: threshold exactly equal (as a python float) to one of the fp32 values -> strict '<' in fp32
        thr = float(np.sort(v)[5])
        cfg = {"enabled": True, "threshold": thr, "layers_to_classify": [lid]}
        cls_out[f"{lid}@edge"] = {"threshold": thr, "result": jsonable(RefClassifier(o.vae, cfg).classify(data4, 4))}

    # ---- nudger: factor/cap variants on a fresh gamma vector
    nud = {}
    pname = "encoder.down_blocks.0.resnets.0.norm1.weight"
    gamma0 = o.vae.get_parameter(pname).detach().clone()
    arrays["nudger/gamma0"] = gamma0.numpy()
    idx = [0, 2, 5, 15, 127]
    for factor, cap, strat in [(1.05, 1.5, "gentle_nudge_groupnorm_scale"), (1.10, 1.5, "gentle_nudge_groupnorm_scale"),
                               (1.20, 1.0, "gentle_nudge_groupnorm_scale"), (1.10, 1.5, "reset_groupnorm_scale")]:
        with torch.no_grad():
            o.vae.get_parameter(pname).copy_(gamma0)
        h = RefNudger(o.vae, {"enabled": True, "strategy": strat, "nudge_factor": factor, "max_scale_value": cap,
                              "intervention_interval": 20})
        res = {"layer": {"param_name_scale": pname, "inactive_channel_indices": idx}}
        h.intervene(res, 10)   # not due: no-op
        assert torch.equal(o.vae.get_parameter(pname).detach(), gamma0)
        h.intervene(res, 20)
        h.intervene(res, 40)   # compounding second nudge
        key = f"{strat}/{factor}/{cap}"
        arrays[f"nudger/{key}"] = o.vae.get_parameter(pname).detach().clone().numpy()
        nud[key] = {"indices": idx, "num_nudges_applied": h.num_nudges_applied}
    with torch.no_grad():
        o.vae.get_parameter(pname).copy_(gamma0)

    # ---- dead weights: three modes on FRESH synthetic weights (portable: no training involved) + planted zeros
    o2 = vo.OracleWrapper(seed=42)
    with torch.no_grad():
        w = o2.vae.get_parameter("decoder.conv_out.weight")
        w.view(-1)[:100] = 0.0
        w.view(-1)[100:200] = 5e-6
        o2.vae.get_parameter("encoder.mid_block.attentions.0.to_k.bias").zero_()  # all-zero tensor: the mean<1e-9 branch
    dead = {}
    for mode in ("threshold", "percent_of_mean", "both"):
        t = RefDNT(TARGET_CLASSES, ["encoder.conv_in.weight"], threshold=1e-5, mean_percentage=0.1, dead_type=mode)
        t.track_dead_neurons(o2.vae, 7)
        dead[mode] = {k: v[0][1] for k, v in t.percent_history.items()}
        if mode == "threshold":
            arrays["dead/raw/encoder.conv_in.weight"] = t.weights_history["encoder.conv_in.weight"][0]
    dead["planted"] = {"param": "decoder.conv_out.weight", "zeros": 100, "tiny": 100, "tiny_value": 5e-6}
    dead["reference_kat"] = {"doc": "deadneuron.py:183-202 dummy net, dead_type both thr 1e-5 pct 0.1", "conv1.weight": [0.0, 1 / 216 * 100]}

    with open(os.path.join(HERE, "e2e_r32.json"), "w") as f:
        json.dump(jsonable(e2e), f, indent=1)
    with open(os.path.join(HERE, "tracker.json"), "w") as f:
        json.dump(tracker, f, indent=1)
    with open(os.path.join(HERE, "classifier.json"), "w") as f:
        json.dump(cls_out, f, indent=1)
    with open(os.path.join(HERE, "nudger.json"), "w") as f:
        json.dump(nud, f, indent=1)
    with open(os.path.join(HERE, "deadneuron.json"), "w") as f:
        json.dump(jsonable(dead), f, indent=1)
    np.savez_compressed(os.path.join(HERE, "arrays.npz"), **arrays)
    print("wrote fixtures:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
