"""The two Keras callbacks the reference's training script installs
(/root/reference model_train.py:259-263), restated with tf.keras 2.10 semantics, plus the
History object `Model.fit` returns (model_train.py:345-357 reads `history.history`).
"""
import numpy as np


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []

    def on_epoch_end(self, epoch, logs):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


class Callback:
    model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


def _better(mode, monitor):
    if mode not in ("auto", "min", "max"):
        mode = "auto"
    if mode == "auto":
        mode = "max" if ("acc" in monitor or monitor.startswith("fmeasure") or "score" in monitor) else "min"
    return mode


class ReduceLROnPlateau(Callback):
    """new_lr = lr * factor once `monitor` has not improved by min_delta for `patience` epochs;
    the wait counter restarts after each reduction and during `cooldown`."""

    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto",
                 min_delta=1e-4, cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor >= 1.0. Got %s" % factor)
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.mode = _better(mode, monitor)
        self.on_train_begin()

    def on_train_begin(self, logs=None):
        self.cooldown_counter = 0
        self.wait = 0
        self.best = np.inf if self.mode == "min" else -np.inf

    def _improved(self, current):
        if self.mode == "min":
            return current < self.best - self.min_delta
        return current > self.best + self.min_delta

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs["lr"] = float(self.model.optimizer.learning_rate)
        current = logs.get(self.monitor)
        if current is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._improved(current):
            self.best = current
            self.wait = 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = float(self.model.optimizer.learning_rate)
                if old > np.float32(self.min_lr):
                    new = max(old * self.factor, self.min_lr)
                    self.model.optimizer.learning_rate = new
                    if self.verbose > 0:
                        print("\nEpoch %d: ReduceLROnPlateau reducing learning rate to %s." % (epoch + 1, new))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class ModelCheckpoint(Callback):
    """Saves the model at the end of each epoch under `filepath.format(epoch=epoch+1, **logs)`,
    or only when `monitor` improved if save_best_only."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, mode="auto"):
        self.filepath, self.monitor, self.verbose = str(filepath), monitor, verbose
        self.save_best_only = save_best_only
        self.mode = _better(mode, monitor)
        self.best = np.inf if self.mode == "min" else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            current = logs.get(self.monitor)
            if current is None:
                return
            better = current < self.best if self.mode == "min" else current > self.best
            if not better:
                return
            if self.verbose > 0:
                print("\nEpoch %d: %s improved from %.5f to %.5f, saving model to %s"
                      % (epoch + 1, self.monitor, self.best, current, path))
            self.best = current
        if getattr(self.model, "is_chief", True):
            self.model.save(path)
