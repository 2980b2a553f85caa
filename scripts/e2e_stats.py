"""Print end-to-end parity statistics of the HIP engine against the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import miseg_amd
from miseg_amd import engine, netspec
from oracle import nets, ops_torch, post, losses
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_e2e import CONFIGS, make_inputs

cfgs = CONFIGS + [("resnet34", "U-Net", (256, 256), 1, 3)]
if "--full" in sys.argv:      # the headline configuration
    cfgs = [("resnet34", "U-Net", (1920, 2560), 1, 3)]
for enc, dec, (h, w), n, classes in cfgs:
    g = netspec.build_graph(enc, dec, (h, w, 3), classes)
    weights = nets.init_weights(nets.build_spec(enc, dec, classes), seed=7, randomize_bn=True)
    x = make_inputs(n, h, w)
    act = "softmax" if classes > 1 else "sigmoid"
    taps = {}
    want, logits = nets.forward(ops_torch, enc, dec, weights, x, activation=act, taps=taps, return_logits=True)
    want = ops_torch.to_numpy(want); logits = ops_torch.to_numpy(logits)
    eng = engine.Engine(g, weights, batch=n)
    got = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    worst = ("", 0)
    for name, view in eng.taps.items():
        if name in taps:
            a = view.torch().float().cpu().numpy(); b = taps[name]
            err = np.abs(a - b).max() / (np.abs(b).max() + 1e-6)
            if err > worst[1]: worst = (name, err)
    line = "%s %s %dx%d n%d c%d: max|dp|=%.4f  worst tap %s %.4f logits std %.3f" % (enc, dec, h, w, n, classes, np.abs(got - want).max(), worst[0], worst[1], logits.std())
    if classes > 1:
        m_got = eng.mask.cpu().numpy()
        m_want = np.stack([post.collapse_probabilities(p) for p in want])
        agree = (m_got == m_want).all(axis=-1)
        top2 = np.sort(want, axis=-1); margin = top2[..., -1] - top2[..., -2]
        line += "  agree %.5f  frac(margin<0.01)=%.4f  agree(margin>0.03)=%.5f" % (
            agree.mean(), (margin < 0.01).mean(), agree[margin > 0.03].mean())
        rng = np.random.default_rng(5)
        ids = np.repeat(np.repeat(rng.integers(0, classes, (n, h // 16, w // 16)), 16, 1), 16, 2)
        gt = np.stack([post.one_hot_encode(i[..., None], classes) for i in ids])
        line += "  dF1 %.2e dIoU %.2e" % (abs(losses.f1_score(gt, got) - losses.f1_score(gt, want)),
                                           abs(losses.iou_score(gt, got) - losses.iou_score(gt, want)))
    print(line, flush=True)
