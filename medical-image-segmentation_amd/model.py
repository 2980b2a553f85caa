"""Keras-like `Model` object over the HIP engine: the surface the reference's
callers use on what `make_model` / `load_model` return
(/root/reference utils/model.py:118,135-141,228-234,272-276;
utils/predict.py:70,159; utils/evaluate.py:127; model_train.py:281-329).
"""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

from . import lib as L
from . import losses as LS
from . import netspec
from . import weights as W

CONTAINER_VERSION = 1


class Adam:
    """Hyper-parameter holder with the Keras 2.10 defaults (epsilon 1e-7)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon

    def get_config(self):
        return {"name": "Adam", "learning_rate": self.learning_rate, "beta_1": self.beta_1,
                "beta_2": self.beta_2, "epsilon": self.epsilon}


class DeviceArray:
    """Result of a forward pass: a device tensor that indexes like an eager
    tensor and converts with `.numpy()` (utils/predict.py:159 does
    `model(batch, training=False)[0].numpy()`)."""

    def __init__(self, t):
        self.tensor = t

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, idx):
        return DeviceArray(self.tensor[idx])

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def __array__(self, dtype=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a


class LayerView:
    def __init__(self, model, layer):
        self._m, self._l = model, layer
        self.name = layer.name

    @property
    def output(self):
        return self._l.out

    @property
    def output_shape(self):
        return self._l.output_shape

    def get_config(self):
        return dict(self._l.config, name=self._l.name)

    def get_weights(self):
        return [self._m.weights[self.name][w] for w, _, _ in self._l.weights]

    def set_weights(self, arrays):
        for (w, shape, _), a in zip(self._l.weights, arrays):
            a = np.asarray(a, np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError("layer %s weight %s: expected %r, got %r" % (self.name, w, shape, a.shape))
            self._m.weights[self.name][w] = a
        self._m._invalidate()

    def count_params(self):
        return self._l.count_params()


class Model:
    def __init__(self, graph, weights=None, seed=None, device=None):
        if device is None:
            # the process's current GPU: under one process per GPU (model_train.py / predict.py with
            # several --gpu entries) every rank has selected its own card before it builds a model
            device = "cuda:%d" % torch.cuda.current_device() if torch.cuda.is_available() else "cuda:0"
        self.graph = graph
        self.weights = weights if weights is not None else W.init_weights(graph, seed=seed)
        self._name = graph.name
        self.device = device
        self.optimizer = None
        self.loss = None
        self.metrics = []
        self.encoder_freeze = False
        self._engines = {}
        self._storage = "fp16"

    # ---- identity -----------------------------------------------------------
    @property
    def name(self):
        return self._name

    @property
    def input_shape(self):
        h, w, c = self.graph.input_shape
        return (None, h, w, c)

    @property
    def output_shape(self):
        h, w, _ = self.graph.input_shape
        return (None, h, w, self.graph.classes)

    @property
    def layers(self):
        return [LayerView(self, l) for l in self.graph.layers.values()]

    def get_layer(self, name=None, index=None):
        if name is not None:
            return LayerView(self, self.graph.layers[name])
        return self.layers[index]

    def count_params(self):
        return self.graph.count_params()

    def summary(self, print_fn=print):
        print_fn('Model: "%s"' % self.name)
        print_fn("%-44s %-26s %12s" % ("Layer (type)", "Output Shape", "Param #"))
        print_fn("=" * 84)
        for l in self.graph.layers.values():
            print_fn("%-44s %-26s %12d" % ("%s (%s)" % (l.name, l.cls), str(l.output_shape),
                                           l.count_params()))
        total = self.count_params()
        non_trainable = sum(int(np.prod(s)) for _, w, s, _ in self.graph.weight_spec()
                            if w in ("moving_mean", "moving_variance"))
        print_fn("=" * 84)
        print_fn("Total params: %d" % total)
        print_fn("Trainable params: %d" % (total - non_trainable))
        print_fn("Non-trainable params: %d" % non_trainable)

    # ---- weights / config -----------------------------------------------------
    def _invalidate(self):
        self._engines = {}
        self._storage = "fp16"          # new weights: the range check of __call__ starts over

    def get_weights(self):
        return W.to_list(self.graph, self.weights)

    def set_weights(self, arrays):
        self.weights = W.from_list(self.graph, arrays)
        self._invalidate()

    def get_config(self):
        g = self.graph
        layers = [{"name": l.name, "class_name": l.cls, "config": dict(l.config, name=l.name),
                   "inbound_nodes": [[[n, 0, 0, {}] for n in l.inbound]] if l.inbound else []}
                  for l in g.layers.values()]
        return {"name": self.name, "layers": layers,
                "input_layers": [[layers[0]["name"], 0, 0]],
                "output_layers": [[layers[-1]["name"], 0, 0]],
                "miseg": {"encoder": g.encoder, "decoder": g.decoder, "classes": g.classes,
                          "activation": g.activation, "temperature": g.temperature}}

    def to_json(self):
        return json.dumps({"class_name": "Functional", "config": self.get_config(),
                           "backend": "miseg_amd", "keras_version": "2.10.0"})

    def compile(self, optimizer=None, loss=None, metrics=None):
        self.optimizer = optimizer if optimizer is not None else Adam()
        self.loss = loss
        flat = []
        for m in (metrics or []):
            flat.extend(m if isinstance(m, (list, tuple)) else [m])
        self.metrics = flat

    # ---- execution -----------------------------------------------------------------
    def _engine(self, batch):
        from . import engine
        e = self._engines.get(batch)
        if e is None:
            self._engines = {}          # one resident plan at a time: buffers are per batch size
            e = engine.Engine(self.graph, self.weights, batch=batch, device=self.device,
                              storage=self._storage)
            self._engines[batch] = e
        return e

    RANGE_CHECK_EVERY = 32

    def _forward(self, batch):
        """Inference forward pass on the engine.  The engine stores activations in fp16, whose range ends at
        65504 (the kernels' conversions saturate there): activations of trained networks stay within a few
        tens, a deep network with untrained weights may not.  The first batch a plan runs, and every
        RANGE_CHECK_EVERY-th after it (one pass over the plan's buffers costs about a fifth of a forward pass,
        so it is not made per batch), is therefore followed by a look at the largest magnitude any of its
        16-bit buffers holds; a plan that reached the end of the range is replaced, from then on, by one with
        bf16 storage (fp32's range, 8 significand bits), and the batch is run again.  New weights
        (`set_weights`, training) start the check over.  What the look cannot see are the tensors the fused
        kernels keep on chip (a DenseNet layer's 128 channels, the C/4-channel tensors of LinkNet's blocks, the
        U-Net tail's 32 channels): they saturate the same way, and their consumers' outputs are inspected."""
        eng = self._engine(batch.shape[0])
        probs = eng.forward(batch)
        count = getattr(eng, "range_batches", 0)
        eng.range_batches = count + 1
        if eng.storage == "fp16" and count % self.RANGE_CHECK_EVERY == 0:
            peak = torch.zeros((), dtype=torch.float32, device=probs.device)
            for t in eng._keep:
                if t.dtype == torch.float16 and t.numel():
                    lo, hi = torch.aminmax(t)
                    peak = torch.maximum(peak, torch.maximum(lo.abs(), hi.abs()).float())
            peak = float(peak)                  # one device -> host read for the whole plan
            if not peak < 65000.0:          # saturated, or NaN
                import warnings
                warnings.warn("activations of %s reach the end of the fp16 range: its inference plans use bf16 "
                              "storage from now on" % self.name)
                self._storage = "bf16"
                self._engines = {}
                eng = self._engine(batch.shape[0])
                probs = eng.forward(batch)
        return eng, probs

    def __call__(self, batch, training=False):
        if isinstance(batch, DeviceArray):
            batch = batch.tensor
        n = batch.shape[0]
        if training:
            # forward on batch statistics, moving averages untouched (no optimisation step)
            if not isinstance(batch, torch.Tensor):
                batch = torch.from_numpy(np.ascontiguousarray(np.asarray(batch, dtype=np.float32)))
            t = self._trainer(n)
            return DeviceArray(t.forward(batch.to(self.device, torch.float32), momentum_update=False).clone())
        _, probs = self._forward(batch)
        return DeviceArray(probs.clone())

    def predict(self, x, batch_size=1, verbose=0):
        out = []
        for i in range(0, len(x), batch_size):
            out.append(self(x[i:i + batch_size], training=False).numpy())
        return np.concatenate(out, axis=0)

    def predict_masks(self, batch, pixel_intensity=127):
        """Forward + collapse_probabilities (utils/utils.py:50-68) on the device:
        returns uint8 (n,H,W,classes)."""
        n = batch.shape[0]
        if pixel_intensity != self._engine(n).collapse_value:
            raise ValueError("the plan was built for pixel_intensity=%d" % self._engine(n).collapse_value)
        eng, _ = self._forward(batch)
        return eng.mask[:n].cpu().numpy()

    def evaluate(self, dataset, return_dict=True, verbose=0, steps=None):
        """Mean over batches of loss and metrics, as Keras logs them.  `dataset`
        yields (images, one-hot masks)."""
        loss = self._loss_object()
        names = ["loss"] + [getattr(m, "name", getattr(m, "__name__", str(m))) for m in self.metrics]
        totals = np.zeros(len(names))
        count = 0
        for images, masks in dataset:
            pr = self(np.asarray(images, np.float32) if not isinstance(images, torch.Tensor) else images)
            s, pixels = LS.class_sums(masks, pr.tensor, focal=getattr(loss, "needs_focal", False))
            vals = [loss.from_sums(s, pixels)] + [m.from_sums(s, pixels) for m in self.metrics]
            totals += np.asarray(vals)
            count += 1
            if steps is not None and count >= steps:
                break
        if count == 0:
            raise ValueError("evaluate() got an empty dataset")
        means = totals / count
        return dict(zip(names, map(float, means))) if return_dict else list(map(float, means))

    # ---- training ----------------------------------------------------------------------
    def _loss_object(self):
        """The compiled loss as an object of losses.py (strings are looked up by name)."""
        loss = self.loss if self.loss is not None else LS.cce_dice_loss
        if isinstance(loss, str):
            loss = LS.BY_NAME.get(loss, LS.cce_dice_loss)
        return loss

    def _loss_name(self):
        loss = self.loss if self.loss is not None else LS.cce_dice_loss
        return loss if isinstance(loss, str) else getattr(loss, "name", getattr(loss, "__name__", str(loss)))

    def _trainer(self, batch):
        from . import trainer
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        t = getattr(self, "_train_state", None)
        if t is None:
            opt = self.optimizer if self.optimizer is not None else Adam()
            t = trainer.Trainer(self.graph, self.weights, batch, device=self.device,
                                learning_rate=opt.learning_rate, beta_1=opt.beta_1, beta_2=opt.beta_2,
                                epsilon=opt.epsilon, loss=self._loss_name(), world_size=world,
                                freeze_encoder=self.encoder_freeze)
            self._train_state = t
        elif t.n != batch:
            # parameters, optimiser state and BatchNorm scratch do not depend on the batch size and
            # activations are allocated per step: a short last batch of a dataset costs nothing
            t.set_batch(batch)
        return t

    @property
    def is_chief(self):
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def train_on_batch(self, x, y, return_dict=True):
        """One optimisation step on the LOCAL batch of this rank (with torch.distributed
        initialised: the per-replica batch of a mirrored strategy; gradients are averaged over
        ranks).  Returns the batch's loss and metrics."""
        dev = torch.device(self.device)

        def prep(a):
            if not isinstance(a, torch.Tensor):
                a = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
            return a.to(dev, torch.float32).contiguous()
        sums, pixels = self._train_step_async(prep(x), prep(y))
        logs = self._logs_from_sums(sums.cpu().numpy(), pixels)
        return logs if return_dict else list(logs.values())

    def _train_step_async(self, x, y):
        """One optimisation step without a host synchronisation: returns the step's per-class sums
        as a DEVICE tensor (a copy: the trainer reuses its buffer) and the pixel count."""
        t = self._trainer(x.shape[0])
        t.lr = float(self.optimizer.learning_rate) if self.optimizer is not None else t.lr
        sums, pixels = t.train_step(x, y)
        return sums.clone(), pixels

    def _logs_from_sums(self, sums, pixels):
        s = np.asarray(sums).reshape(-1, self.graph.classes)
        loss = self._loss_object()
        logs = OrderedDict([("loss", loss.from_sums(s, pixels))])
        for m in self.metrics:
            logs[getattr(m, "name", getattr(m, "__name__", str(m)))] = m.from_sums(s, pixels)
        return logs

    def sync_from_trainer(self, collective=True):
        """Copy the trained parameters (and moving statistics, averaged over ranks) back into
        `self.weights`, which inference, evaluate() and save() read.  With `collective` (the
        default) this is a COLLECTIVE call when torch.distributed is initialised: every rank must
        make it (fit() does, at the end of every epoch).  collective=False only copies, and is
        what single-rank callers such as save() use."""
        t = getattr(self, "_train_state", None)
        if t is None:
            return
        if collective and getattr(t, "dirty", True):
            t.average_moving_stats()
        if getattr(t, "dirty", True):
            self.weights = t.get_weights()
            self._engines = {}
            t.dirty = False

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None,
            validation_data=None, steps_per_epoch=None, initial_epoch=0, validation_steps=None,
            shuffle=True):
        """Keras-style training loop (the call at model_train.py:284-329).  `x` is an iterable of
        (images, one-hot masks) batches (a repeating dataset needs steps_per_epoch) or, with `y`,
        arrays cut into batches of `batch_size`.  With torch.distributed initialised every rank
        calls fit() on its own shard of the data: one process per GPU, gradients all-reduced on
        RCCL during the backward pass.  Returns a History."""
        from . import callbacks as CB
        if self.optimizer is None:
            self.compile(optimizer=Adam(), loss=self.loss, metrics=self.metrics)
        history = CB.History()
        cbs = list(callbacks or [])
        for cb in cbs:
            cb.set_model(self)
            cb.on_train_begin()

        def batches():
            if y is not None:
                n = len(x)
                bs = batch_size or 32
                idx = np.random.permutation(n) if shuffle else np.arange(n)
                for i in range(0, n - bs + 1 if n >= bs else n, bs):
                    j = idx[i:i + bs]
                    yield x[j], y[j]
            else:
                for b in x:
                    yield b[0], b[1]
        it = None
        for epoch in range(initial_epoch, epochs):
            for cb in cbs:
                cb.on_epoch_begin(epoch)
            if it is None or y is not None:
                it = iter(batches())
            totals, count, fresh = OrderedDict(), 0, False
            pending = []        # (device sums, pixels) of the epoch's steps: read back once, at its end
            while steps_per_epoch is None or count < steps_per_epoch:
                try:
                    bx, by = next(it)
                    fresh = False
                except StopIteration:
                    if steps_per_epoch is None or fresh:
                        it = None               # the epoch is one pass (or the dataset is empty)
                        break
                    it, fresh = iter(batches()), True   # a finite dataset is simply walked again
                    continue
                dev = torch.device(self.device)
                bx, by = (a if isinstance(a, torch.Tensor) else
                          torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))) for a in (bx, by))
                pending.append(self._train_step_async(bx.to(dev, torch.float32).contiguous(),
                                                      by.to(dev, torch.float32).contiguous()))
                count += 1
            if count == 0:
                raise ValueError("fit() got an empty dataset")
            # one device -> host transfer per epoch instead of a synchronisation per step
            host = torch.stack([p[0] for p in pending]).cpu().numpy()
            for row, (_, pixels) in zip(host, pending):
                for k, v in self._logs_from_sums(row, pixels).items():
                    totals[k] = totals.get(k, 0.0) + v
            logs = OrderedDict((k, v / count) for k, v in totals.items())
            self._reduce_logs(logs)
            self.sync_from_trainer()
            if validation_data is not None:
                val = self.evaluate(validation_data, steps=validation_steps)
                for k, v in val.items():
                    logs["val_" + k] = v
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            history.on_epoch_end(epoch, logs)
            if verbose and self.is_chief:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in logs.items()))
        for cb in cbs:
            cb.on_train_end()
        return history

    def _reduce_logs(self, logs):
        """Average the epoch's training logs over ranks (each rank saw its own shard)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        keys = list(logs)
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([logs[k] for k in keys], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        t /= dist.get_world_size()
        for k, v in zip(keys, t.tolist()):
            logs[k] = v

    # ---- persistence -----------------------------------------------------------------
    def save(self, path):
        """`*.h5` / `*.hdf5`: a Keras HDF5 model file (keras_h5.save_keras_h5), which is what the
        reference's ModelCheckpoint writes (model_train.py:187-190,262).  Any other name: a
        single-file numpy container (.npz layout): the JSON config plus every weight under
        '<layer>/<weight>:0'.  Both are written atomically and read back by `load_model`.

        save() issues no collective, so one rank alone (the chief, from ModelCheckpoint) may call
        it: fit() has already averaged the moving statistics over the ranks on every rank at the
        end of the epoch.  Called in the middle of an epoch of a data-parallel run it writes this
        rank's own moving statistics."""
        self.sync_from_trainer(collective=False)
        if str(path).endswith((".h5", ".hdf5")):
            from . import keras_h5
            keras_h5.save_keras_h5(self, path)
            return
        arrays = {"__config__": np.frombuffer(self.to_json().encode(), dtype=np.uint8),
                  "__version__": np.asarray([CONTAINER_VERSION])}
        for l, w, _, _ in self.graph.weight_spec():
            arrays["%s/%s:0" % (l, w)] = self.weights[l][w]
        tmp = str(path) + ".tmp"
        with open(tmp, "wb") as f:
            np.savez(f, **arrays)
        os.replace(tmp, str(path))


def model_from_json(text, custom_objects=None):
    cfg = json.loads(text)["config"]
    meta = cfg.get("miseg")
    if meta is None:
        raise ValueError("this JSON was not written by miseg_amd.Model.to_json(): no 'miseg' block")
    shape = cfg["layers"][0]["config"]["batch_input_shape"][1:]
    g = netspec.build_graph(meta["encoder"], meta["decoder"], tuple(shape), meta["classes"],
                            activation=meta["activation"], temperature=meta["temperature"],
                            name=cfg.get("name"))
    return Model(g, seed=0)


def load_container(path):
    with np.load(str(path), allow_pickle=False) as z:
        text = bytes(z["__config__"]).decode()
        model = model_from_json(text)
        weights = OrderedDict()
        for l, w, shape, _ in model.graph.weight_spec():
            a = z["%s/%s:0" % (l, w)]
            if tuple(a.shape) != tuple(shape):
                raise ValueError("%s/%s: stored shape %r does not match the graph's %r" % (l, w, a.shape, shape))
            weights.setdefault(l, OrderedDict())[w] = a.astype(np.float32)
    model.weights = weights
    return model
