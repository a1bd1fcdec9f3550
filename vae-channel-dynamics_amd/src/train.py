"""SDXL-VAE fine-tuning with channel-dynamics tracking on MI355X.

Drop-in for the reference entry point (reference src/train.py:100-464): same `--config_path` CLI,
same YAML keys, same output files (config.yaml, chkpt-N/, final_model/{model.safetensors,
optimizer.bin, scheduler.bin, random_states_0.pkl, vae/}, tracked_activation_stats.csv,
intervention_history.csv).  Launch one process per GPU:

    python vae-channel-dynamics_amd/src/train.py --config_path <yaml>                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           vae-channel-dynamics_amd/src/train.py --config_path <yaml>                       # 8 GPUs, RCCL/xGMI
    (`accelerate launch ...` sets the same RANK/WORLD_SIZE environment and works too.)

What differs from the reference, on purpose:
  * the step body (train.py:283-306) is HipTrainer.train_step: HIP kernels + RCCL bucketed gradient mean,
    no per-step host sync (the reference does 3 gathers + 3 .item() every step, train.py:292-297);
    the logging scalars are summed on the device and read once per log interval / epoch.
  * tracker statistics are averaged over ranks and classification + intervention run on EVERY rank,
    so replicas stay identical (the reference nudges rank 0 only and never re-broadcasts).
  * wandb / tensorboard are optional: when unavailable, metrics go to <output_dir>/metrics.jsonl.
  * plots are not produced (reporting only); every CSV the plots were made from is still written.
"""
import argparse
import csv
import json
import logging
import math
import os
import pickle
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

from utils.config_utils import load_config
from utils.logging_utils import setup_logging
from data_utils import load_and_preprocess_dataset, create_dataloader
from models.sdxl_vae_wrapper import SDXLVAEWrapper
from tracking.monitor import ActivityMonitor
from tracking.deadneuron import DeadNeuronTracker
from classification.classifier import RegionClassifier
from intervention.nudger import InterventionHandler
from vaehip.trainer import HipTrainer
from vaehip.dp import allreduce_mean_

setup_logging()
logger = logging.getLogger(__name__)

target_layer_classes = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear, torch.nn.GroupNorm)


def parse_args():
    p = argparse.ArgumentParser(description="Train/Fine-tune SDXL VAE with channel dynamics analysis.")
    p.add_argument("--config_path", type=str, required=True, help="Path to the YAML configuration file for the experiment.")
    return p.parse_args()


class MetricLogger:
    """wandb / tensorboard when importable and requested, always a JSONL file on the main process."""

    def __init__(self, report_to, output_dir, logging_dir, project, run_name, config, entity, is_main):
        self.is_main = is_main
        self.wandb = None
        self.tb = None
        self.fh = None
        if not is_main:
            return
        os.makedirs(output_dir, exist_ok=True)
        self.fh = open(os.path.join(output_dir, "metrics.jsonl"), "a")
        if report_to in ("wandb", "all"):
            try:
                import wandb
                wandb.init(project=project, name=run_name, config=config, dir=output_dir, entity=entity)
                self.wandb = wandb
            except Exception as e:
                logger.error(f"W&B init failed: {e}. No W&B logging.")
        if report_to in ("tensorboard", "all"):
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.tb = SummaryWriter(logging_dir)
            except Exception as e:
                logger.warning(f"tensorboard unavailable ({e}); metrics.jsonl only.")

    def log(self, metrics: dict, step: int):
        if not self.is_main:
            return
        clean = {k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in metrics.items()}
        self.fh.write(json.dumps({"step": step, **clean}) + "\n")
        self.fh.flush()
        if self.wandb is not None:
            self.wandb.log(clean, step=step)
        if self.tb is not None:
            for k, v in clean.items():
                if isinstance(v, float):
                    self.tb.add_scalar(k, v, step)

    def close(self):
        if self.fh:
            self.fh.close()
        if self.wandb is not None:
            self.wandb.finish()
        if self.tb is not None:
            self.tb.close()


def with_last(iterable):
    """yields (item, is_last): accelerate's dataloader wrapper knows the end of a pass one batch ahead
    (gradient_state.end_of_dataloader), which is what forces an optimizer update on a partial accumulation window"""
    it = iter(iterable)
    try:
        prev = next(it)
    except StopIteration:
        return
    for cur in it:
        yield prev, False
        prev = cur
    yield prev, True


_CTRL_GROUP = None


def all_ranks_ok(ok: bool, world: int) -> bool:
    """control-plane agreement on a host flag (MIN over ranks) through a gloo group: a rank whose batch failed to
    decode must not leave its peers alone in the gradient exchange.  CPU tensors, so no device synchronisation."""
    global _CTRL_GROUP
    if world <= 1:
        return ok
    if _CTRL_GROUP is None:
        _CTRL_GROUP = dist.new_group(backend="gloo")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_CTRL_GROUP)
    return bool(flag.item())


def agreed_batches(loader, world: int):
    """yields (batch, every rank's batch is valid, last of the pass).  The agreement on batch k+1 (all_ranks_ok's int32 MIN
    all-reduce on the gloo control group) is STARTED when the look-ahead of `with_last` fetches that batch and awaited one step
    later, so it runs behind batch k's host work instead of in front of every step (blocking: 0.36 ms per call at 2 ranks,
    0.9 ms at 8 on an 8-core host, tools/ctrl_allreduce_cost.py).  Every rank issues the same collectives in the same order."""
    global _CTRL_GROUP

    def start(b):
        ok = valid_batch(b)
        if world <= 1:
            return ok, None, None
        global _CTRL_GROUP
        if _CTRL_GROUP is None:
            _CTRL_GROUP = dist.new_group(backend="gloo")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        return ok, flag, dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_CTRL_GROUP, async_op=True)

    def finish(tok):
        ok, flag, work = tok
        if work is None:
            return ok
        work.wait()
        return bool(flag.item())

    it = iter(loader)
    try:
        cur = next(it)
    except StopIteration:
        return
    tok = ntok = start(cur)
    try:
        for nxt in it:
            ntok = start(nxt)
            yield cur, finish(tok), False
            cur, tok = nxt, ntok
        yield cur, finish(tok), True
    finally:  # a consumer that stops early (max_train_steps) leaves one look-ahead all-reduce in flight: complete it here,
        if ntok is not None and ntok[2] is not None:  # before the caller's barrier / destroy_process_group
            ntok[2].wait()


def valid_batch(batch) -> bool:
    pv = batch.get("pixel_values") if batch else None  # guards the reference's None-batch crash (data_utils.py:215)
    return pv is not None and pv.ndim == 4 and pv.shape[0] > 0


def save_state(path: str, wrapper, trainer, rank: int = 0):
    """file layout of accelerate.save_state (accelerate/utils/constants.py:20-31) used by train.py:358-362,392-405."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in wrapper.state_dict().items()}
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    torch.save(trainer.optimizer.state_dict(), os.path.join(path, "optimizer.bin"))
    torch.save(trainer.lr_scheduler.state_dict(), os.path.join(path, "scheduler.bin"))
    states = {"step": trainer.global_step, "random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
              "torch_manual_seed": torch.get_rng_state(),
              "torch_cuda_manual_seed": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None}
    with open(os.path.join(path, f"random_states_{rank}.pkl"), "wb") as f:
        pickle.dump(states, f)


def run_validation(trainer: HipTrainer, val_dataloader, kl_weight: float, global_step: int, mlog: MetricLogger, device, world: int):
    """train.py:53-97: eval forward with mode(), SUM-reduced MSE and kl().sum(), per-sample averages.
    Hooks/trackers stay attached, so validation forwards enter the monitor buffer exactly as in the reference."""
    logger.info(f"--- Running Validation for Global Step: {global_step} ---")
    trainer.wrapper.eval()
    sums = torch.zeros(3, device=device, dtype=torch.float64)  # rec_sum, kl_sum, samples
    for batch, ok_everywhere, _ in agreed_batches(val_dataloader, world):
        if not ok_everywhere:  # every rank skips together (tracker buffers stay aligned)
            logger.warning("Validation: Invalid batch data, skipping.")
            continue
        pv = batch["pixel_values"]
        r = trainer.eval_step(pv.to(device, dtype=torch.float32, non_blocking=True))
        sums[0] += r["rec_sum"].double()
        sums[1] += r["kl_sum"].double()
        sums[2] += pv.shape[0]
    if world > 1:
        dist.all_reduce(sums)
    rec_sum, kl_sum, n = sums.tolist()
    avg_rec = rec_sum / n if n > 0 else 0
    avg_kl = kl_sum / n if n > 0 else 0
    avg = avg_rec + kl_weight * avg_kl
    logger.info(f"  Avg Validation Loss (Total): {avg:.4e}, Rec: {avg_rec:.4e}, KL: {avg_kl:.4e}; validated on {int(n)} samples.")
    m = {"validation/avg_total_loss": avg, "validation/avg_reconstruction_loss": avg_rec, "validation/avg_kl_divergence": avg_kl}
    mlog.log(m, global_step)
    trainer.wrapper.train()
    return m


def main():
    args = parse_args()
    config = load_config(args.config_path)
    run_name = config.get("run_name", "vae_run")
    threshold_dn = float(config.get("threshold", 1e-8))
    mean_percentage_dn = float(config.get("mean_percentage", .01))
    dead_type_dn = config.get("dead_type", "threshold")
    output_dir = os.path.join(config.get("output_dir", "./results"), run_name)
    logging_dir = os.path.join(output_dir, "logs")
    logging_cfg = config.get("logging", {})
    report_to = logging_cfg.get("report_to", "tensorboard")
    training_cfg = config.get("training", {})
    mixed_precision = training_cfg.get("mixed_precision", "no")
    if mixed_precision not in ("no", "bf16"):
        raise NotImplementedError(f"training.mixed_precision={mixed_precision!r}: this path has 'no' (exact fp32 MFMA) and "
                                  "'bf16' (bf16 MFMA products, fp32 accumulate, fp32 master weights and statistics)")
    grad_accum = int(training_cfg.get("gradient_accumulation_steps", 1))
    # not a key of the reference's YAMLs (it never checkpoints): "decoder" re-runs the decoder forward before its backward
    # instead of keeping its activations (BASELINE config 5)
    grad_ckpt = training_cfg.get("gradient_checkpointing", False)
    if grad_ckpt not in (False, None, "decoder", True):
        raise NotImplementedError(f"training.gradient_checkpointing={grad_ckpt!r}: only 'decoder' exists")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this trainer runs on MI355X only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=device)
    is_main = rank == 0
    logger.info(f"rank {rank}/{world} on {device}; running experiment: {run_name}")
    if config.get("seed") is not None:
        seed = int(config["seed"])
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
    if is_main:
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "config.yaml"), "w") as f:
            yaml.dump(config, f, default_flow_style=False)
    mlog = MetricLogger(report_to, output_dir, logging_dir, config.get("project_name", "vae_project"), run_name, config,
                        logging_cfg.get("entity", None), is_main)
    if world > 1:
        dist.barrier()

    model_cfg = config.get("model", {})
    # parameters stay fp32 in every mode (the reference casts them to bf16 when mixed_precision is bf16,
    # train.py:147-154: no master copy -- SURVEY 3.4; here bf16 is a compute mode only)
    vae_wrapper = SDXLVAEWrapper(pretrained_model_name_or_path=model_cfg.get("pretrained_vae_name", "stabilityai/sdxl-vae"),
                                 torch_dtype=None, device=device)
    data_cfg = config.get("data", {})
    bs = data_cfg.get("batch_size", 4)
    # data.gpu_preprocess (not a key of the reference): workers only decode; Resize / CenterCrop / Normalize run on the GPU
    gpu_pre = None
    if data_cfg.get("gpu_preprocess", False):
        from vaehip.preprocess import GpuPreprocessor
        gpu_pre = GpuPreprocessor(data_cfg.get("resolution", 256), device)
    train_dataset = load_and_preprocess_dataset(
        gpu_preprocess=gpu_pre is not None,
        dataset_name=data_cfg.get("dataset_name"), dataset_config_name=data_cfg.get("dataset_config_name", None),
        image_column=data_cfg.get("image_column", "image"), resolution=data_cfg.get("resolution", 256),
        max_samples=data_cfg.get("max_samples", None), split=data_cfg.get("train_split_name", "train"))
    train_dataloader = create_dataloader(train_dataset, batch_size=bs, num_workers=data_cfg.get("num_workers", 0),
                                         shuffle=True, rank=rank, world_size=world, seed=int(config.get("seed") or 0),
                                         gpu_preprocess=gpu_pre)
    val_dataloader = None
    if data_cfg.get("do_validation", False):
        try:
            vds = load_and_preprocess_dataset(
                gpu_preprocess=gpu_pre is not None,
                dataset_name=data_cfg.get("validation_dataset_name", data_cfg.get("dataset_name")),
                dataset_config_name=data_cfg.get("validation_dataset_config_name", data_cfg.get("dataset_config_name", None)),
                image_column=data_cfg.get("image_column", "image"), resolution=data_cfg.get("resolution", 256),
                max_samples=data_cfg.get("validation_max_samples", None), split=data_cfg.get("validation_split_name", "validation"))
            val_dataloader = create_dataloader(vds, batch_size=data_cfg.get("validation_batch_size", bs),
                                               num_workers=data_cfg.get("num_workers", 0), shuffle=False, rank=rank, world_size=world,
                                               gpu_preprocess=gpu_pre)
        except Exception as e:
            logger.error(f"Failed to load validation data: {e}. Disabling validation.")
            data_cfg["do_validation"] = False

    # schedule arithmetic exactly as train.py:188-195 (max_train_steps ignores the world size)
    n_train = len(train_dataset) if hasattr(train_dataset, "__len__") else None
    steps_per_epoch = math.ceil(n_train / bs / grad_accum) if n_train else training_cfg.get("max_steps_per_epoch_iterable", 10000)
    num_train_epochs = int(training_cfg.get("num_train_epochs", 1))
    max_train_steps = num_train_epochs * steps_per_epoch
    kl_weight = float(training_cfg.get("kl_weight", 1e-6))
    max_grad_norm = float(training_cfg.get("max_grad_norm", 1.0))
    trainer = HipTrainer(
        vae_wrapper, lr=float(training_cfg.get("learning_rate", 1e-5)),
        betas=(training_cfg.get("adam_beta1", 0.9), training_cfg.get("adam_beta2", 0.999)),
        eps=training_cfg.get("adam_epsilon", 1e-08), weight_decay=training_cfg.get("adam_weight_decay", 1e-2),
        max_grad_norm=max_grad_norm, kl_weight=kl_weight, lr_warmup_steps=int(training_cfg.get("lr_warmup_steps", 100)),
        max_train_steps=max_train_steps, scheduler_steps_per_update=world,  # accelerate steps the scheduler `world` times
        mixed_precision=mixed_precision, gradient_accumulation_steps=grad_accum, checkpoint_decoder=bool(grad_ckpt))

    core_vae = vae_wrapper.vae
    dnt_cfg = config.get("dead_neuron_tracking", {})
    dnt = DeadNeuronTracker(target_layer_classes, dnt_cfg.get("target_layer_names_for_raw_weights", []), threshold_dn,
                            mean_percentage_dn, dead_type_dn) if dnt_cfg.get("enabled", False) else None
    monitor_cfg = config.get("tracking", {})
    monitor = ActivityMonitor(vae_wrapper, monitor_cfg) if monitor_cfg.get("enabled", False) else None
    cls_cfg = config.get("classification", {})
    classifier = RegionClassifier(model=core_vae, config=cls_cfg) if cls_cfg.get("enabled", False) else None
    int_cfg = config.get("intervention", {})
    intervention = InterventionHandler(model=core_vae, config=int_cfg) if int_cfg.get("enabled", False) else None
    if config.get("logit_lens", {}).get("enabled", False):
        logger.info("logit_lens.enabled: visualisation is out of scope of this build (in the reference's train loop it "
                    "never matches a layer key anyway, SURVEY.md 2a #10); skipped.")

    log_interval = int(logging_cfg.get("log_interval", 10))
    save_interval = int(config.get("saving", {}).get("save_interval_steps", 500))
    ckpt_prefix = config.get("saving", {}).get("checkpoint_dir_prefix", "chkpt")
    track_interval = int(monitor_cfg.get("track_interval", 100)) if monitor else -1
    dnt_interval = int(dnt_cfg.get("track_interval", 100)) if dnt else -1
    val_epochs = int(training_cfg.get("validation_epochs", 0))
    val_steps = int(training_cfg.get("validation_steps", 0))

    logger.info(f"***** Running training: {num_train_epochs} epochs x {steps_per_epoch} steps, batch {bs}/GPU x {world} GPU *****")
    global_step = 0
    t_start = time.time()
    for epoch in range(num_train_epochs):
        vae_wrapper.train()
        if hasattr(train_dataloader.sampler, "set_epoch"):
            train_dataloader.sampler.set_epoch(epoch)
        epoch_sums = torch.zeros(3, device=device, dtype=torch.float64)  # mse, kl, total (this rank)
        steps_in_epoch = 0
        sc = torch.zeros(3, device=device)
        for batch, ok_everywhere, last_batch in agreed_batches(train_dataloader, world):
            if not ok_everywhere:  # all ranks skip the batch, or none does
                if last_batch and trainer.pending_micro_batches:
                    trainer.flush()  # the pass ends on a skipped batch: update with what has accumulated
                else:
                    continue
            else:
                pv = batch["pixel_values"]
                # the last batch of a pass always ends in an optimizer update (accelerate.accumulate, train.py:286)
                res = trainer.train_step(pv.to(device, dtype=torch.float32, non_blocking=True), end_of_dataloader=last_batch)
                sc = res["scalars"]
                epoch_sums += sc.double()
                steps_in_epoch += 1
            if not trainer.sync_gradients:  # micro-batch of an accumulated update (train.py:286,300)
                continue
            global_step += 1
            activity_logs = {}
            if monitor and global_step % track_interval == 0:
                activity_logs = monitor.step(global_step)
            classification = {}
            if classifier and monitor and global_step % track_interval == 0:
                tracked = monitor.get_data_for_step(global_step)
                if tracked:
                    classification = classifier.classify(tracked, global_step)
                if not classification:
                    logger.info(f"Step {global_step}: Classifier found no inactive channels.")
            if intervention and global_step % int_cfg.get("intervention_interval", 200) == 0:
                if classification:
                    intervention.intervene(classification, global_step)
                    inactive_total = sum(len(v["inactive_channel_indices"]) for v in classification.values())
                    nudged = intervention.num_nudges_applied
                    mlog.log({"inactive_channels": inactive_total, "nudged_scales": nudged}, global_step)
                    if is_main:
                        with open(os.path.join(output_dir, "intervention_history.csv"), "a") as fh:
                            fh.write(f"{global_step},{inactive_total},{nudged}\n")
                else:
                    logger.info(f"Step {global_step}: Intervention due, but no regions classified.")
            if global_step % log_interval == 0:
                step_loss = allreduce_mean_(sc[2:3].clone()).item()  # the only host sync of a logging step
                if is_main:
                    mlog.log({"train_loss_step": step_loss, "lr": trainer.lr_scheduler.get_last_lr()[0], "epoch_current": epoch,
                              **activity_logs}, global_step)
                    logger.info(f"step {global_step}/{max_train_steps} loss {step_loss:.4e} lr {trainer.lr_scheduler.get_last_lr()[0]:.3e} "
                                f"({(time.time() - t_start) / global_step * 1e3:.0f} ms/step)")
            if dnt and global_step % dnt_interval == 0:
                dnt.track_dead_neurons(core_vae, global_step)
            if global_step % save_interval == 0 and is_main:
                d = os.path.join(output_dir, f"{ckpt_prefix}-{global_step}")
                save_state(d, vae_wrapper, trainer)
                logger.info(f"Saved periodic training state to {d}")
            if val_dataloader and data_cfg.get("do_validation", False) and val_steps > 0 and global_step % val_steps == 0:
                run_validation(trainer, val_dataloader, kl_weight, global_step, mlog, device, world)
            if global_step >= max_train_steps:
                break
        if world > 1:
            dist.all_reduce(epoch_sums)
            epoch_sums /= world
        es = (epoch_sums / max(steps_in_epoch, 1)).tolist()
        nan = float("nan")
        mlog.log({"train/epoch_avg_loss": es[2] if steps_in_epoch else nan, "train/epoch_avg_rec_loss": es[0] if steps_in_epoch else nan,
                  "train/epoch_avg_kl_loss": es[1] if steps_in_epoch else nan, "epoch_completed": epoch}, global_step)
        logger.info(f"Epoch {epoch} completed. Avg Train Loss: {es[2] if steps_in_epoch else nan:.4e}")
        if val_dataloader and data_cfg.get("do_validation", False) and val_epochs > 0 and (epoch + 1) % val_epochs == 0 and val_steps <= 0:
            run_validation(trainer, val_dataloader, kl_weight, global_step, mlog, device, world)
        if global_step >= max_train_steps:
            logger.info("Reached max_train_steps.")
            break
    if world > 1:
        dist.barrier()
    logger.info("Training finished.")

    if is_main:
        final_dir = os.path.join(output_dir, "final_model")
        save_state(final_dir, vae_wrapper, trainer)
        core_vae.save_pretrained(os.path.join(final_dir, "vae"))  # what evaluate.py loads (evaluate.py:91-102)
        logger.info(f"Final training state and unwrapped VAE saved under {final_dir}")
        if monitor:
            recs = monitor.export_all_processed_data_to_records()
            if recs:
                path = os.path.join(output_dir, "tracked_activation_stats.csv")
                with open(path, "w", newline="") as f:
                    wr = csv.DictWriter(f, fieldnames=list(recs[0].keys()))
                    wr.writeheader()
                    wr.writerows(recs)
                logger.info(f"Saved activation stats to {path}")
        if dnt and dnt.percent_history:
            path = os.path.join(output_dir, "dead_neuron_percentage_history.csv")
            with open(path, "w", newline="") as f:
                wr = csv.writer(f)
                wr.writerow(["layer", "step", "percentage"])
                for name, hist in dnt.percent_history.items():
                    for step, pct in hist:
                        wr.writerow([name, step, pct])
            logger.info(f"Saved dead-weight percentages to {path}")
    mlog.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as e:
        logging.getLogger(__name__).error(f"Unhandled exception in main: {e}", exc_info=True)
        sys.exit(1)
