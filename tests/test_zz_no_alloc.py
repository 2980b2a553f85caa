"""miseg_bn_stats_fwd_noalloc (the entry point that allocates nothing: its partial sums live in an
array built into the library) against miseg_bn_stats_fwd with a caller-supplied workspace, through
the C ABI, on two calls in a row, the second with a smaller input (fewer pixels and fewer
channels), so that whatever the first call left in the shared array must not show in the second.

Bar: the two entry points run the same kernels on the same data; the only difference between two
runs is the order of the fp64 atomics that merge fp32 per-thread partial sums, so the fp32 results
must agree to 1e-6 relative (they are normally bit-identical).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, L, dev, fn, x, c, stride, gamma, beta, mm0, mv0, with_ws):
    pixels = x.shape[0]
    mm, mv = mm0.clone(), mv0.clone()
    mean, var, scale, shift = (torch.zeros(c, device=dev) for _ in range(4))
    args = [x.data_ptr(), pixels, c, stride, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
            0.99, 1e-3, mean.data_ptr(), var.data_ptr(), scale.data_ptr(), shift.data_ptr()]
    if with_ws:
        ws = torch.full((2 * c,), 7.0, dtype=torch.float64, device=dev)   # the call must clear it
        args.append(ws.data_ptr())
    args.append(None)
    L.check(fn(*args))
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (mean, var, scale, shift, mm, mv)]


def test_bn_stats_noalloc_matches_workspace_entry_point():
    import miseg_amd
    from miseg_amd import lib as L
    lib = L.load()
    L.check(lib.miseg_device_check())
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    # (pixels, channels, pixel stride): the second call is smaller in both extents
    for pixels, c, stride in ((2 * 37 * 53, 264, 272), (1 * 19 * 23, 40, 48)):
        x = torch.from_numpy(rng.normal(0.2, 1.3, (pixels, stride)).astype(np.float32)).to(torch.bfloat16).to(dev)
        gamma = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(dev)
        beta = torch.from_numpy(rng.normal(0, 0.5, c).astype(np.float32)).to(dev)
        mm0 = torch.from_numpy(rng.normal(0, 1, c).astype(np.float32)).to(dev)
        mv0 = torch.from_numpy(rng.uniform(0.5, 2, c).astype(np.float32)).to(dev)
        want = _run(lib, L, dev, lib.miseg_bn_stats_fwd, x, c, stride, gamma, beta, mm0, mv0, True)
        got = _run(lib, L, dev, lib.miseg_bn_stats_fwd_noalloc, x, c, stride, gamma, beta, mm0, mv0, False)
        for name, g, w in zip(("mean", "var", "scale", "shift", "moving_mean", "moving_var"), got, want):
            np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-7, err_msg="%s, C=%d" % (name, c))
        # and against numpy, so that two equally wrong entry points cannot pass
        xr = x[:, :c].float().cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(got[0], xr.mean(0), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got[1], xr.var(0), rtol=1e-5)


def test_bn_stats_noalloc_rejects_too_many_channels():
    import miseg_amd
    from miseg_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    c = 4104
    x = torch.zeros((4, c), dtype=torch.bfloat16, device=dev)
    f = torch.zeros(c, device=dev)
    rc = lib.miseg_bn_stats_fwd_noalloc(x.data_ptr(), 4, c, c, f.data_ptr(), f.data_ptr(), None, None, 0.99, 1e-3,
                                        f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), None)
    assert rc == L.E_INVALID_ARG
    assert b"4096" in lib.miseg_last_error()
