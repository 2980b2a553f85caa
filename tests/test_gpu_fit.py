"""Model.fit on the HIP path: the surface model_train.py:259-329 uses (History keys, validation,
ReduceLROnPlateau / ModelCheckpoint, encoder_freeze) and the data-parallel step."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(n, h, w, c=3, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.random((n, h, w, 3), dtype=np.float32)
    lab = (x[..., 0] > 0.5).astype(np.int64) + (x[..., 1] > 0.7).astype(np.int64)
    return x, np.eye(c, dtype=np.float32)[lab]


def _batches(x, y, bs):
    return [(x[i:i + bs], y[i:i + bs]) for i in range(0, len(x), bs)]


def test_fit_history_callbacks_and_checkpoint(pkg, tmp_path):
    sys.path.insert(0, ROOT)
    from utils import model as um
    from miseg_amd.callbacks import ModelCheckpoint, ReduceLROnPlateau
    m = um.make_model("resnet18", "U-Net", (64, 96, 3), 3, 1e-3, um.CUSTOM_OBJECTS[
        "categorical_crossentropy_plus_dice_loss"], um.METRICS, seed=1)
    x, y = _data(8, 64, 96)
    xv, yv = _data(2, 64, 96, seed=9)
    ckpt = os.path.join(str(tmp_path), "m_e{epoch:03d}_l{loss:.4f}_vl{val_loss:.4f}.h5")
    cbs = [ReduceLROnPlateau(monitor="val_f1-score", factor=0.1, min_delta=1e-3, min_lr=1e-8, patience=1,
                             mode="max"),
           ModelCheckpoint(ckpt, monitor="val_f1-score", mode="max", save_best_only=False)]
    h = m.fit(_batches(x, y, 2), epochs=3, steps_per_epoch=6, validation_data=_batches(xv, yv, 1),
              callbacks=cbs, verbose=0)
    assert set(h.history) == {"loss", "f1-score", "iou_score", "val_loss", "val_f1-score", "val_iou_score", "lr"}
    assert h.epoch == [0, 1, 2] and all(len(v) == 3 for v in h.history.values())
    assert h.history["loss"][-1] < h.history["loss"][0]
    files = sorted(os.listdir(str(tmp_path)))
    assert len(files) == 3 and files[0].startswith("m_e001_l")
    # the checkpoint of the last epoch reproduces the validation numbers fit() logged
    m2 = um.load_model(os.path.join(str(tmp_path), files[-1]), input_shape=(64, 96, 3))
    m2.compile(loss=m.loss, metrics=m.metrics)
    ev = m2.evaluate(_batches(xv, yv, 1))
    assert abs(ev["loss"] - h.history["val_loss"][-1]) < 1e-6
    assert abs(ev["f1-score"] - h.history["val_f1-score"][-1]) < 1e-6
    # resume: initial_epoch continues the epoch numbering
    h2 = m.fit(_batches(x, y, 2), epochs=4, steps_per_epoch=2, initial_epoch=3, verbose=0)
    assert h2.epoch == [3]


def test_encoder_freeze_keeps_backbone(pkg):
    sys.path.insert(0, ROOT)
    from utils import model as um
    m = um.make_model("resnet18", "Linknet", (64, 64, 3), 3, 1e-3, um.CUSTOM_OBJECTS[
        "categorical_crossentropy_plus_dice_loss"], um.METRICS, encoder_freeze=True, seed=2)
    before = {k: {n: v.copy() for n, v in d.items()} for k, d in m.weights.items()}
    x, y = _data(4, 64, 64)
    m.fit(_batches(x, y, 2), epochs=1, steps_per_epoch=3, verbose=0)
    g = m.graph
    names = list(g.layers)
    enc = set(names[:names.index(g.encoder_output) + 1])
    changed_dec = 0
    for k, d in m.weights.items():
        for n, v in d.items():
            same = np.array_equal(v, before[k][n])
            if k in enc:
                assert same, (k, n)
            elif n in ("kernel", "gamma", "beta"):
                changed_dec += (not same)
    assert changed_dec > 10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out_dir, enc="resnet18", dec="U-Net", hw=(64, 96), bucket_mb=8):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import miseg_amd  # noqa: F401
    from miseg_amd import netspec, trainer
    from oracle import nets
    w = nets.init_weights(nets.build_spec(enc, dec, 3), 0, True)
    g = netspec.build_graph(enc, dec, (hw[0], hw[1], 3), 3, "softmax", None)
    x, y = _data(2, hw[0], hw[1], seed=4)
    tr = trainer.Trainer(g, w, 1, world_size=world, bucket_mb=bucket_mb)
    xd = torch.from_numpy(x[rank:rank + 1]).cuda()
    yd = torch.from_numpy(y[rank:rank + 1]).cuda()
    tr.forward(xd)
    tr.loss_and_metrics(yd)
    tr.backward(loss_scale=1.0 / world)
    tr.buckets.wait()
    torch.cuda.synchronize()
    torch.save({"G": tr.G.cpu(), "buckets": tr.buckets.launched}, os.path.join(out_dir, "dp%d.pt" % rank))
    dist.destroy_process_group()


# the DenseNet-169 FPN is the data-parallel training configuration of the reference's README; its backward
# writes several layers' gradients from one fused call (head, final_stage, the folded finest level), so the
# buckets must not leave before the last of them.  2 MB buckets: a dozen of them for these networks.
@pytest.mark.parametrize("enc,dec,hw,bucket_mb", [("resnet18", "U-Net", (64, 96), 8),
                                                  ("densenet169", "FPN", (64, 64), 2),
                                                  ("densenet121", "Linknet", (64, 64), 2)])
def test_data_parallel_gradients_two_ranks_one_gpu(pkg, tmp_path, enc, dec, hw, bucket_mb):
    """Two ranks (sharing the one GPU of the test box, gloo staging through the host) each take
    one image; the all-reduced gradient equals the mean of the two single-rank gradients."""
    import torch
    import torch.multiprocessing as mp
    from miseg_amd import netspec, trainer
    from oracle import nets
    world = 2
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path), enc, dec, hw, bucket_mb), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "dp0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "dp1.pt"))
    assert torch.equal(r0["G"], r1["G"])
    assert len(r0["buckets"]) >= 3                          # several buckets, not one per layer
    assert sum(hi - lo for lo, hi in r0["buckets"]) == r0["G"].numel()
    w = nets.init_weights(nets.build_spec(enc, dec, 3), 0, True)
    g = netspec.build_graph(enc, dec, (hw[0], hw[1], 3), 3, "softmax", None)
    x, y = _data(2, hw[0], hw[1], seed=4)
    tr = trainer.Trainer(g, w, 1)
    total = None
    for r in range(world):
        tr.forward(torch.from_numpy(x[r:r + 1]).cuda(), momentum_update=False)
        tr.loss_and_metrics(torch.from_numpy(y[r:r + 1]).cuda())
        tr.backward(loss_scale=1.0 / world)
        total = tr.G.clone() if total is None else total + tr.G
    ref = total.cpu()
    err = float((r0["G"] - ref).norm() / ref.norm())
    assert err < 1e-3, err            # float atomics in the weight-gradient kernel reorder sums


def _fit_ckpt_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from utils import model as um
    from miseg_amd.callbacks import ModelCheckpoint, ReduceLROnPlateau
    m = um.make_model("resnet18", "U-Net", (64, 64, 3), 3, 1e-3, um.CUSTOM_OBJECTS[
        "categorical_crossentropy_plus_dice_loss"], um.METRICS, seed=1)
    x, y = _data(8, 64, 64, seed=3)
    xv, yv = _data(2, 64, 64, seed=9)
    ckpt = os.path.join(out_dir, "dp_e{epoch:03d}.h5")
    cbs = [ReduceLROnPlateau(monitor="val_f1-score", factor=0.5, min_delta=1e-3, min_lr=1e-8, patience=1, mode="max"),
           ModelCheckpoint(ckpt, monitor="val_f1-score", mode="max", save_best_only=False)]
    # each rank trains on its own half of the data; only the chief writes checkpoints
    h = m.fit(_batches(x[rank::world], y[rank::world], 2), epochs=2, steps_per_epoch=2,
              validation_data=_batches(xv, yv, 1), callbacks=cbs, verbose=0)
    m.sync_from_trainer()                 # collective: every rank
    flat = np.concatenate([v.ravel() for d in m.weights.values() for v in d.values()])
    torch.save({"history": h.history, "weights": torch.from_numpy(flat)}, os.path.join(out_dir, "fit%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_with_checkpoint_two_ranks_stays_in_step(pkg, tmp_path):
    """Two data-parallel ranks run fit() with a ModelCheckpoint over two epochs: the chief alone
    saves, so saving must not issue a collective the other rank never joins (that would pair the
    chief's all-reduce with the other rank's next gradient bucket: a hang or corrupted gradients).
    Afterwards both ranks hold the same parameters and moving statistics and logged the same
    epoch metrics."""
    import torch
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_fit_ckpt_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "fit0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "fit1.pt"))
    assert torch.equal(r0["weights"], r1["weights"])
    assert r0["history"] == r1["history"] and len(r0["history"]["val_loss"]) == 2
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".h5"))
    assert files == ["dp_e001.h5", "dp_e002.h5"]
