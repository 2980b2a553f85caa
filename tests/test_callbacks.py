"""Keras-semantics checks of the two callbacks model_train.py:259-263 installs."""
import os

import pytest


class _Opt:
    learning_rate = 1e-3


class _FakeModel:
    def __init__(self):
        self.optimizer = _Opt()
        self.saved = []
        self.is_chief = True

    def save(self, path):
        self.saved.append(path)


def test_reduce_lr_on_plateau_max_mode(pkg):
    from miseg_amd.callbacks import ReduceLROnPlateau
    m = _FakeModel()
    cb = ReduceLROnPlateau(monitor="val_f1-score", factor=0.5, min_delta=1e-3, min_lr=1e-8, patience=2,
                           mode="max")
    cb.set_model(m)
    cb.on_train_begin()
    seq = [0.50, 0.60, 0.6005, 0.6004, 0.59, 0.58, 0.70]
    lrs = []
    for e, v in enumerate(seq):
        logs = {"val_f1-score": v}
        cb.on_epoch_end(e, logs)
        lrs.append((logs["lr"], m.optimizer.learning_rate))
    # epochs 2,3 do not beat 0.60 by 1e-3 -> reduced after epoch 3; wait restarts; 4,5 -> reduced again
    assert [round(a[1], 8) for a in lrs] == [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 2.5e-4, 2.5e-4]
    assert lrs[3][0] == 1e-3                   # 'lr' logged is the rate the epoch ran with


def test_reduce_lr_respects_min_lr_and_factor_check(pkg):
    from miseg_amd.callbacks import ReduceLROnPlateau
    with pytest.raises(ValueError):
        ReduceLROnPlateau(factor=1.0)
    m = _FakeModel()
    m.optimizer.learning_rate = 2e-8
    cb = ReduceLROnPlateau(monitor="val_loss", factor=0.1, patience=1, min_lr=1e-8)
    cb.set_model(m)
    cb.on_train_begin()
    for e in range(4):
        cb.on_epoch_end(e, {"val_loss": 1.0})
    assert m.optimizer.learning_rate == 1e-8


def test_model_checkpoint_names_and_best_only(pkg, tmp_path):
    from miseg_amd.callbacks import ModelCheckpoint
    m = _FakeModel()
    pat = os.path.join(str(tmp_path), "AgNOR_e{epoch:03d}_l{loss:.4f}_vl{val_loss:.4f}.h5")
    cb = ModelCheckpoint(pat, monitor="val_f1-score", mode="max", save_best_only=False)
    cb.set_model(m)
    cb.on_epoch_end(0, {"loss": 0.51234, "val_loss": 0.49999, "val_f1-score": 0.7})
    assert os.path.basename(m.saved[0]) == "AgNOR_e001_l0.5123_vl0.5000.h5"
    m2 = _FakeModel()
    cb = ModelCheckpoint(os.path.join(str(tmp_path), "best.h5"), monitor="val_f1-score", mode="max",
                         save_best_only=True)
    cb.set_model(m2)
    for e, v in enumerate([0.5, 0.4, 0.6, 0.6]):
        cb.on_epoch_end(e, {"val_f1-score": v})
    assert len(m2.saved) == 2
