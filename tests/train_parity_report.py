"""Per-layer parameter-gradient agreement of one HIP training step with the CPU oracle, fp32 and
with bf16 storage emulated (oracle/ops_bf16.py).  Not collected by pytest: run by hand,
  python tests/train_parity_report.py resnet18 U-Net 128 192 2
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(enc, dec, h, w, n):
    import train_selfcheck as sc
    from oracle import train, ops_bf16
    tr, weights, x, gt = sc.run(enc, dec, h, w, n)
    gw = sc.grads_as_weights(tr)
    out = {}
    for label, ops in (("fp32", None), ("bf16", ops_bf16)):
        loss, probs, grads, stats = train.loss_and_grads(enc, dec, weights, x, gt, ops=ops)
        rows = []
        for l in grads:
            for k in grads[l]:
                ref = grads[l][k]
                nr = np.linalg.norm(ref)
                if nr == 0:
                    continue
                rows.append((l, k, float(np.linalg.norm(gw[l][k] - ref) / nr)))
        a = np.concatenate([gw[l][k].ravel() for l in grads for k in grads[l]])
        b = np.concatenate([grads[l][k].ravel() for l in grads for k in grads[l]])
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        out[label] = (loss, rows, cos, float(np.linalg.norm(a - b) / np.linalg.norm(b)))
    return out


if __name__ == "__main__":
    enc, dec, h, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    r = report(enc, dec, h, w, n)
    f32 = dict(((l, k), e) for l, k, e in r["fp32"][1])
    print("%s %s %dx%d n=%d  loss fp32 %.6f bf16 %.6f" % (enc, dec, h, w, n, r["fp32"][0], r["bf16"][0]))
    print("whole gradient: cosine vs fp32 %.4f (rel err %.3f), vs bf16 emulation %.4f (rel err %.3f)"
          % (r["fp32"][2], r["fp32"][3], r["bf16"][2], r["bf16"][3]))
    for l, k, e in r["bf16"][1]:
        print("%-36s %-8s vs fp32 %.4f   vs bf16 %.4f" % (l, k, f32[(l, k)], e))
