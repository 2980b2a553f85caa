"""Per-call timing of libmiseg.so entry points with HIP events recorded on the stream each call
is launched on (the training step is driven from Python, call by call, so there is no plan to
replay): the source of the `roofline` object of bench.py's training leg.

Algorithmic work per call: convolutions and weight gradients report their multiply-accumulates
and operand bytes through `note()`; for the bandwidth-bound layers the bytes follow from the
arguments (every operand read once, every result written once, 2 bytes per bf16 element).
"""
import collections

import torch


def _bytes_bn_apply(a):        # x, dy, dx, pixels, C, ..., addend at index 15
    return a[3] * a[4] * 2 * (3 + (1 if a[15] else 0))


BYTES = {
    "miseg_bn_relu_bwd_apply": _bytes_bn_apply,
    "miseg_bn_relu_bwd_reduce": lambda a: a[2] * a[3] * 4,
    "miseg_scale_shift_relu_fwd": lambda a: a[4] * a[5] * 4,
    "miseg_bn_stats_fwd": lambda a: a[1] * a[2] * 2,
    "miseg_copy_channels": lambda a: a[3] * a[4] * 2 * (2 + (1 if a[1] else 0)),
    "miseg_channel_sums": lambda a: a[1] * a[2] * 2,
    "miseg_maxpool3x3s2_fwd": lambda a: a[4] * a[5] * a[6] * a[7] * 2 * 1.25,
    "miseg_maxpool3x3s2_bwd": lambda a: a[3] * a[4] * a[5] * a[6] * 2 * (2.25 + (1 if a[10] else 0)),
    "miseg_avgpool2x2_fwd": lambda a: a[2] * a[3] * a[4] * a[5] * 2 * 1.25,
    "miseg_avgpool2x2_bwd": lambda a: a[2] * a[3] * a[4] * a[5] * 2 * 1.25,
    "miseg_loss_sums": lambda a: a[3] * a[4] * 8,
    "miseg_loss_sums_focal": lambda a: a[3] * a[4] * 8,
    "miseg_loss_grad_ex": lambda a: a[4] * (a[5] * 8 + a[6] * 2),
    "miseg_adam_step": lambda a: a[5] * (4 * 7 + 2),
    "miseg_prep_input_indicator": lambda a: a[4] * a[5] * a[6] * (12 + 8),
    "miseg_input_stats_fwd": lambda a: a[1] * 12,
}


class TimedLib:
    """Proxy of the ctypes library object: transparent while `on` is False; while True, every
    asynchronous entry point is bracketed by two events on its own stream."""

    def __init__(self, lib):
        self._lib = lib
        self.on = False
        self.records = []
        self.tag = ""                  # set by the caller: which layer the next calls belong to
        self._note = None

    def note(self, kernel, macs=0, nbytes=0):
        """Label and algorithmic work of the next call (convolutions, weight gradients)."""
        if self.on:
            self._note = (kernel, macs, nbytes)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not self.on or not name.startswith("miseg_") or fn.restype is None or \
                name in ("miseg_last_error", "miseg_device_check", "miseg_version", "miseg_conv_kernel_name"):
            return fn

        def call(*args):
            ptr = args[-1]
            stream = torch.cuda.ExternalStream(ptr) if ptr else torch.cuda.default_stream()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = fn(*args)
            e1.record(stream)
            note, self._note = self._note, None
            if note is None:
                f = BYTES.get(name)
                note = (name.replace("miseg_", "") + "_kernel", 0, f([a or 0 for a in args]) if f else 0)
            self.records.append((name, note, e0, e1, self.tag))
            return rc
        return call

    def summary(self):
        """{kernel: {"launches", "ms", "flops", "bytes"}} of the recorded calls (synchronises)."""
        torch.cuda.synchronize()
        out = collections.OrderedDict()
        for name, (kernel, macs, nbytes), e0, e1, _ in self.records:
            g = out.setdefault(kernel, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "entry": name})
            g["launches"] += 1
            g["ms"] += e0.elapsed_time(e1)
            g["flops"] += 2.0 * macs
            g["bytes"] += float(nbytes)
        return out

    def calls(self):
        """[(layer tag, kernel, ms, flops, bytes)] of the recorded calls in launch order (synchronises)."""
        torch.cuda.synchronize()
        return [(tag, kernel, e0.elapsed_time(e1), 2.0 * macs, float(nbytes))
                for _, (kernel, macs, nbytes), e0, e1, tag in self.records]
