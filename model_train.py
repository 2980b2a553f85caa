#!/usr/bin/env python3
"""Train a segmentation model: command line and bookkeeping of the reference's training script
(/root/reference model_train.py:103-543) on the MI355X path.

  python model_train.py --encoder resnet18 --decoder U-Net --dataset DATA --epochs 10 \\
         --batch-size 4 --steps 50 --height 1920 --width 2560

`--gpu 0,1,2,3` (several devices) re-launches this script under torch.distributed.run with one
process per GPU (data parallel: each rank takes every world-th sample and the gradients are
all-reduced on RCCL); `--gpu -1` (the reference's CPU mode) is refused: this path has no CPU
fallback.  Checkpoints are Keras HDF5 files named as the reference names them.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time
from pathlib import Path


def show_train_config(cfg, start_time=None, end_time=None, duration=None):
    if start_time:
        print("\nTraining start: %s" % start_time)
    if end_time:
        print("Training end: %s" % end_time)
    if duration:
        print("Duration: %s" % duration)
    rows = [("Directory", "directory"), ("Encoder", "encoder"), ("Decoder", "decoder"),
            ("Loss function", "loss_function"), ("Initial learning rate", "initial_learning_rate"),
            ("Learning rate factor", "learning_rate_factor"), ("Classes", "classes"), ("Epochs", "epochs"),
            ("Patience", "patience"), ("Batch size", "batch_size"), ("Steps per epoch", "steps_per_epoch"),
            ("Input shape", "input_shape"), ("Augmentation", "augmentation"), ("One hot encoded", "one_hot_encoded"),
            ("Encoder weights freeze", "encoder_freeze"), ("GPU(s)", "gpu"), ("Metrics", "metrics"),
            ("Save all weights", "save_all"), ("Seed", "seed")]
    for label, key in rows:
        print("  - %s: %s" % (label, cfg[key]))
    if cfg["model_name"]:
        print("  - Model name: %s" % cfg["model_name"])
    if cfg["resume"]:
        print("  - Resume model: %s" % cfg["resume"])
        print("  - Resume epoch: %s" % cfg["resume_epoch"])
    print("  - Dataset:")
    for label, key in (("Train", "train"), ("Validation", "validation"), ("Test", "test")):
        print("    - %s: %s (%s samples)" % (label, cfg[key + "_dataset"], cfg[key + "_samples"]))
    print()


def update_best_model_and_metrics(cfg, best_model, models_metrics):
    if "best_model" not in cfg or best_model["f1-score"] > cfg["best_model"]["f1-score"]:
        cfg["best_model"] = best_model
    if "test_metrics" in cfg:
        for k, old in cfg["test_metrics"].items():
            old.extend(models_metrics[k][len(old):])
    else:
        cfg["test_metrics"] = models_metrics
    return cfg


def train(encoder, decoder, dataset, loss, learning_rate, learning_rate_factor, classes, epochs, patience,
          batch_size, steps_per_epoch, height, width, augmentation, rgb, encoder_freeze, save_all, gpu, model_name,
          description, seed, resume, resume_epoch):
    import numpy as np
    import torch
    import torch.distributed as dist
    from utils.data import list_files, load_dataset
    from utils.evaluate import evaluate
    from utils.model import CUSTOM_OBJECTS, METRICS, load_model, make_model
    from utils.utils import add_time_delta, get_duration, optimize_models_storage, plot_metrics
    from miseg_amd.model import Adam
    from miseg_amd.callbacks import ModelCheckpoint, ReduceLROnPlateau

    distributed = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = int(os.environ.get("RANK", "0"))
    if distributed:
        # a GPU listed twice (--gpu 0,0) carries two ranks
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group("nccl")
    chief = rank == 0
    if seed:
        np.random.seed(seed)
        torch.manual_seed(seed)
    input_shape = (height, width, 3) if rgb else (height, width)
    one_hot = classes > 1
    loss_function = {"dice": CUSTOM_OBJECTS["categorical_crossentropy_plus_dice_loss"],
                     "focal": CUSTOM_OBJECTS["focal_loss_plus_dice_loss"],
                     "categorical": CUSTOM_OBJECTS["categorical_crossentropy"]}[loss]
    metrics = list(METRICS)

    if resume:
        ckpt_dir = Path(resume).parent
    else:
        stamp = time.strftime("%Y%m%d%H%M%S")
        if distributed:                                   # every rank must agree on the directory
            box = [stamp]
            dist.broadcast_object_list(box, src=0)
            stamp = box[0]
        ckpt_dir = Path("checkpoints").joinpath(stamp)
        ckpt_dir.mkdir(exist_ok=True, parents=True)
    ckpt_pattern = str(ckpt_dir.joinpath(model_name + ("_e{epoch:03d}_l{loss:.4f}_vl{val_loss:.4f}.h5" if save_all
                                                        else ".h5")))
    paths = {k: Path(dataset).joinpath(k) for k in ("train", "validation", "test")}
    per_rank = max(1, batch_size // (dist.get_world_size() if distributed else 1))
    train_ds = load_dataset(str(paths["train"]), batch_size=per_rank, shape=(height, width), classes=classes,
                            mask_one_hot_encoded=one_hot, repeat=True, shuffle=True, augment=augmentation)
    val_ds = load_dataset(str(paths["validation"]), batch_size=1, shape=(height, width), classes=classes,
                          mask_one_hot_encoded=one_hot, repeat=False, shuffle=True, augment=False, rank=0, world=1)

    cfg_path = str(ckpt_dir.joinpath("train_config_%s.json" % ckpt_dir.name))
    cfg = {
        "directory": ckpt_dir.name, "encoder": encoder, "decoder": decoder,
        "loss_function": getattr(loss_function, "name", getattr(loss_function, "__name__", str(loss_function))),
        "initial_learning_rate": learning_rate, "learning_rate_factor": learning_rate_factor, "classes": classes,
        "epochs": epochs, "patience": patience, "batch_size": batch_size, "steps_per_epoch": steps_per_epoch,
        "input_shape": input_shape, "augmentation": augmentation, "one_hot_encoded": one_hot,
        "encoder_freeze": encoder_freeze, "gpu": str(gpu),
        "metrics": [m if isinstance(m, str) else m.__name__ for m in metrics],
        "train_dataset": str(paths["train"]), "validation_dataset": str(paths["validation"]),
        "test_dataset": str(paths["test"]),
        "train_samples": len(list_files(paths["train"].joinpath("images"), as_numpy=True)),
        "validation_samples": len(list_files(paths["validation"].joinpath("images"), as_numpy=True)),
        "test_samples": len(list_files(paths["test"].joinpath("images"), as_numpy=True)),
        "save_all": save_all, "model_name": model_name, "description": description, "seed": seed,
        "resume": resume, "resume_epoch": resume_epoch,
    }
    if resume:
        with open(cfg_path) as f:
            cfg = {**json.load(f), **cfg}
    elif chief:
        with open(cfg_path, "w") as f:
            json.dump(cfg, f, indent=4)

    callbacks = [ReduceLROnPlateau(monitor="val_f1-score", factor=learning_rate_factor, min_delta=1e-3, min_lr=1e-8,
                                   patience=patience, verbose=1, mode="max"),
                 ModelCheckpoint(ckpt_pattern, monitor="val_f1-score", mode="max", save_best_only=not save_all)]
    start, start_time = time.time(), time.strftime("%x %X")
    try:
        if resume:
            model = load_model(model_path=str(Path(resume)), input_shape=input_shape, loss_function=loss_function,
                               optimizer=Adam(learning_rate=learning_rate))
            # the checkpoint carries per-layer trainable flags and load_model restores them (as Keras'
            # load_model does for the reference, model_train.py:276-281): a run that was started with a
            # frozen encoder stays frozen after --resume; --encoder_freeze can only add the freeze
            model.encoder_freeze = bool(getattr(model, "encoder_freeze", False)) or bool(encoder_freeze)
        else:
            model = make_model(encoder, decoder, input_shape, classes, learning_rate, loss_function, metrics,
                               encoder_freeze, model_name, seed=seed)
        if chief:
            model.summary()
            show_train_config(cfg, start_time)
        history = model.fit(train_ds, epochs=epochs, steps_per_epoch=steps_per_epoch, validation_data=val_ds,
                            initial_epoch=int(resume_epoch) if resume else 0, callbacks=callbacks,
                            verbose=1 if chief else 0)
    except Exception as e:                                # recorded in the config, as the reference does
        print("\nThere was an error during training that caused it to stop: \n%s" % e)
        cfg["error"] = str(e)
        history = None
        if distributed:
            # the other ranks are blocked in the step's gradient all-reduce; going on to the final
            # barrier (a different collective) would hang the job.  Record the error and leave with a
            # failure, so that the launcher tears the whole group down.
            if chief:
                with open(cfg_path, "w") as f:
                    json.dump(cfg, f, indent=4)
            raise
    end, end_time = time.time(), time.strftime("%x %X")
    duration = get_duration(start, end)
    if "duration" in cfg:
        duration = add_time_delta(cfg["duration"], duration)
    cfg["duration"] = duration
    if chief:
        show_train_config(cfg, start_time, end_time, duration)
    if history:
        store = cfg.setdefault("train_metrics", {})
        for k, v in history.history.items():
            store.setdefault(k, []).extend(float(i) for i in v)
    if chief:
        with open(cfg_path, "w") as f:
            json.dump(cfg, f, indent=4)
        if glob.glob(str(ckpt_dir.joinpath("*.h5"))):
            print("\nEvaluate all saved models on test data '%s'" % paths["test"])
            best, per_model = evaluate(models_paths=str(ckpt_dir), images_path=str(paths["test"]), batch_size=1,
                                       classes=classes, one_hot_encoded=one_hot, input_shape=input_shape,
                                       loss_function=loss_function, model_name=model_name)
            if best is not None or per_model is not None:
                cfg = update_best_model_and_metrics(cfg, best, per_model)
                with open(cfg_path, "w") as f:
                    json.dump(cfg, f, indent=4)
                plot_metrics(cfg_path)
                optimize_models_storage(str(ckpt_dir))
        else:
            print("\nSkipping evaluation, no models were found at `%s`." % ckpt_dir)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return cfg


def main():
    p = argparse.ArgumentParser(description="Train a segmentation model.")
    p.add_argument("--encoder", type=str, help="The feature extractor of the model.")
    p.add_argument("--decoder", type=str, help="The decoder of the model.")
    p.add_argument("--dataset", type=str, help="Directory with 'train', 'validation' and 'test', each holding "
                                               "'images' and 'masks'.")
    p.add_argument("--loss", default="dice", type=str, help="dice | focal | categorical")
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--lr-factor", default=0.75, type=float)
    p.add_argument("--classes", default=3, type=int)
    p.add_argument("--epochs", type=int)
    p.add_argument("--patience", default=10, type=int)
    p.add_argument("--batch-size", type=int)
    p.add_argument("--steps", type=int)
    p.add_argument("--height", type=int)
    p.add_argument("--width", type=int)
    p.add_argument("--augmentation", choices=["true", "false"], default="false", type=str)
    p.add_argument("--rgb", default=True, action="store_true")
    p.add_argument("--encoder-freeze", default=False, action="store_true")
    p.add_argument("--save-all", default=False, action="store_true")
    p.add_argument("--gpu", default="0", type=str, help="GPU index, or several separated by commas.")
    p.add_argument("--name", default="AgNOR", type=str)
    p.add_argument("--description", default="", type=str)
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--resume", default=None, type=str)
    p.add_argument("--resume-epoch", default=None, type=int)
    a = p.parse_args()

    if str(a.gpu).strip() == "-1":
        raise SystemExit("--gpu -1 (CPU) is not available: this implementation runs on MI355X GPUs only")
    gpus = [g for g in str(a.gpu).split(",") if g != ""]
    if len(gpus) > 1 and "RANK" not in os.environ:
        # one process per GPU; the children see the same arguments and find RANK in their environment
        env = dict(os.environ, HIP_VISIBLE_DEVICES=",".join(dict.fromkeys(g.strip() for g in gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(gpus)),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if len(gpus) == 1 and "RANK" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", gpus[0])
    cfg = train(encoder=a.encoder, decoder=a.decoder, dataset=a.dataset, loss=a.loss, learning_rate=a.lr,
          learning_rate_factor=a.lr_factor, classes=a.classes, epochs=a.epochs, patience=a.patience,
          batch_size=a.batch_size, steps_per_epoch=a.steps, height=a.height, width=a.width,
          augmentation=a.augmentation.lower() == "true", rgb=a.rgb, encoder_freeze=a.encoder_freeze,
          save_all=a.save_all, gpu=a.gpu, model_name=a.name, description=a.description, seed=a.seed,
          resume=a.resume, resume_epoch=a.resume_epoch)
    if cfg.get("error"):
        # as the reference, a failed fit() still evaluates what was saved; unlike it, the process
        # reports the failure to its caller
        raise SystemExit(1)


if __name__ == "__main__":
    main()
