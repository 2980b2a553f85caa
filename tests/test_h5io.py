"""Pure-Python HDF5 reader/writer and the Keras model-file layout on top of it
(the files /root/reference utils/model.py:265 loads and model_train.py:262 writes).

Pinning: the reader is checked against `testhdf5_7.4_GLNX86.mat`, a MATLAB v7.3 file (HDF5 written
by the HDF5 C library, with a 512-byte user block) that ships in scipy's test data, whose content
scipy itself can decode from the sibling v5 file `testdouble_7.4_GLNX86.mat`.  The writer is checked
against the reader.  No file written by Keras/h5py exists offline: Keras-compatibility of the
writer's output is "parity unpinned" beyond the HDF5 and Keras layout specifications it follows.
"""
import json
import os

import numpy as np
import pytest

SCIPY_DATA = None
try:
    import scipy.io
    SCIPY_DATA = os.path.join(os.path.dirname(scipy.io.__file__), "matlab", "tests", "data")
except Exception:          # pragma: no cover
    pass


@pytest.mark.skipif(SCIPY_DATA is None or not os.path.isfile(os.path.join(SCIPY_DATA or "", "testhdf5_7.4_GLNX86.mat")),
                    reason="scipy's MATLAB test data is not installed")
def test_reader_against_libhdf5_written_file(pkg):
    from miseg_amd import h5io
    import scipy.io
    f = h5io.File(os.path.join(SCIPY_DATA, "testhdf5_7.4_GLNX86.mat"))
    assert f.keys() == ["testdouble"]
    ds = f["testdouble"]
    assert ds.attrs["MATLAB_class"] == b"double"
    want = scipy.io.loadmat(os.path.join(SCIPY_DATA, "testdouble_7.4_GLNX86.mat"))["testdouble"]
    got = ds.read()
    assert got.dtype == np.float64 and got.shape == want.shape[::-1]      # MATLAB stores column-major
    np.testing.assert_array_equal(got.T, want)


def test_writer_reader_round_trip(pkg, tmp_path):
    from miseg_amd import h5io
    rng = np.random.default_rng(0)
    w = h5io.Writer()
    big_json = json.dumps({"layers": ["layer_%d" % i for i in range(20000)]})
    assert len(big_json) > 200000
    w.attr("/", "model_config", big_json.encode())
    w.attr("/", "keras_version", b"2.10.0")
    w.attr("/", "note", "héllo")
    arrays = {}
    for i in range(40):                                   # more entries than a default 8-entry node
        a = rng.normal(size=(3, 3, i + 1, 2)).astype(np.float32)
        arrays["g/layer_%02d/layer_%02d/kernel:0" % (i, i)] = a
        w.dataset("/g/layer_%02d/layer_%02d/kernel:0" % (i, i), a)
        w.attr("/g/layer_%02d" % i, "weight_names", np.asarray([b"layer_%02d/kernel:0" % i]))
    w.dataset("/ints", np.arange(12, dtype=np.int64).reshape(3, 4))
    w.dataset("/empty", np.zeros((0,), np.float32))
    w.attr("/g", "layer_names", np.asarray([b"layer_%02d" % i for i in range(40)]))
    w.attr("/g", "scalar", np.float64(2.5))
    path = os.path.join(str(tmp_path), "t.h5")
    w.save(path)
    f = h5io.File(path)
    assert f.attrs["model_config"] == big_json
    assert f.attrs["keras_version"] == "2.10.0" and f.attrs["note"] == "héllo"
    assert sorted(f.keys()) == ["empty", "g", "ints"]
    assert len(f["g"].keys()) == 40
    assert [bytes(b) for b in f["g"].attrs["layer_names"]] == [b"layer_%02d" % i for i in range(40)]
    assert float(f["g"].attrs["scalar"]) == 2.5
    got = dict(f["g"].visit_datasets("g/"))
    assert set(got) == set(arrays)
    for k, a in arrays.items():
        np.testing.assert_array_equal(got[k].read(), a)
    np.testing.assert_array_equal(f["ints"].read(), np.arange(12).reshape(3, 4))
    assert f["empty"].read().shape == (0,)
    assert bytes(f["g/layer_07"].attrs["weight_names"][0]) == b"layer_07/kernel:0"
    with pytest.raises(KeyError):
        f["g/nope"]
    with open(os.path.join(str(tmp_path), "bad.h5"), "wb") as fh:
        fh.write(b"not hdf5" * 100)
    with pytest.raises(h5io.H5Error):
        h5io.File(os.path.join(str(tmp_path), "bad.h5"))


@pytest.mark.parametrize("enc,dec,tss", [("resnet18", "U-Net", False), ("resnet34", "Linknet", True),
                                         ("densenet121", "FPN", False), ("resnet50", "U-Net", False),
                                         ("vgg16", "Linknet", False)])
def test_keras_layout_round_trip(pkg, tmp_path, enc, dec, tss):
    from miseg_amd import keras_h5, netspec, h5io
    from miseg_amd.model import Model
    from miseg_amd import weights as W
    g = netspec.build_graph(enc, dec, (64, 96, 3), 3, temperature=0.1 if tss else None, name="AgNOR_test")
    m = Model(g, weights=W.init_weights(g, seed=5, randomize_bn=True))
    path = os.path.join(str(tmp_path), "m.h5")
    keras_h5.save_keras_h5(m, path)
    f = h5io.File(path)
    # Keras layout: layer groups keyed by layer name, datasets by '<layer>/<weight>:0'
    assert f.attrs["backend"] == "tensorflow"
    names = [bytes(b).decode() for k in sorted(a for a in f["model_weights"].attrs if a.startswith("layer_names"))
             for b in f["model_weights"].attrs[k]]
    assert names == list(g.layers)
    some = "conv1/conv" if enc.startswith("densenet") else "block1_conv1" if enc.startswith("vgg") else "conv0"
    assert f["model_weights"][some][some + "/kernel:0"].shape == ((3, 3, 3, 64) if enc.startswith("vgg")
                                                                  else (7, 7, 3, 64))
    m2 = keras_h5.load_keras_h5(path)
    assert m2.name == "AgNOR_test"
    assert (m2.graph.encoder, m2.graph.decoder, m2.graph.classes) == (enc, dec, 3)
    assert m2.graph.temperature == (0.1 if tss else None)
    assert m2.input_shape == (None, 64, 96, 3)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)
    cfg = json.loads(f.attrs["model_config"])
    arch = keras_h5.infer_architecture(cfg)
    assert arch["encoder"] == enc and arch["decoder"] == dec
    # weights-only use: same groups at the root are accepted too
    m3 = Model(g, weights=W.init_weights(g, seed=6))
    keras_h5.load_weights_h5(m3, path)
    np.testing.assert_array_equal(m3.get_weights()[0], m.get_weights()[0])


def test_load_model_sniffs_format_and_reports_mismatch(pkg, tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from utils import model as um
    from miseg_amd import keras_h5, netspec, h5io
    from miseg_amd.model import Model
    g = netspec.build_graph("resnet18", "FPN", (64, 64, 3), 3)
    m = Model(g, seed=1)
    p_h5 = os.path.join(str(tmp_path), "a.h5")
    p_npz = os.path.join(str(tmp_path), "b.h5")            # container under an .h5 name still loads
    m.save(p_h5)
    with open(p_h5, "rb") as fh:
        assert fh.read(8) == h5io.SIGNATURE
    import shutil
    m.save(os.path.join(str(tmp_path), "c.npz"))
    shutil.copy(os.path.join(str(tmp_path), "c.npz"), p_npz)
    for p in (p_h5, p_npz):
        m2 = um.load_model(p, compile=False)
        np.testing.assert_array_equal(m2.get_weights()[5], m.get_weights()[5])
    m3 = um.load_model(p_h5, input_shape=(96, 64, 3), compile=False)
    assert um.get_model_input_shape(m3) == (96, 64, 3)
    # a file whose weights do not fit the recognised architecture is refused with the layer named
    w = h5io.Writer()
    w.attr("/", "model_config", json.dumps(keras_h5.keras_model_config(m)).encode())
    w.dataset("/model_weights/conv0/conv0/kernel:0", np.zeros((7, 7, 3, 32), np.float32))
    w.attr("/model_weights/conv0", "weight_names", np.asarray([b"conv0/kernel:0"]))
    bad = os.path.join(str(tmp_path), "bad.h5")
    w.save(bad)
    with pytest.raises(ValueError, match="bn_data|conv0"):
        um.load_model(bad, compile=False)


def test_frozen_encoder_survives_the_round_trip(pkg, tmp_path):
    """encoder_freeze is stored as the backbone layers' `trainable` flags (as Keras stores it) and read back."""
    from miseg_amd import keras_h5, netspec
    from miseg_amd.model import Model
    from miseg_amd import weights as W
    import json
    from miseg_amd import h5io
    g = netspec.build_graph("resnet18", "Linknet", (64, 64, 3), 3)
    m = Model(g, weights=W.init_weights(g, seed=2))
    m.encoder_freeze = True
    path = os.path.join(str(tmp_path), "frozen.h5")
    keras_h5.save_keras_h5(m, path)
    cfg = json.loads(keras_h5._text(h5io.File(path).attrs["model_config"]))["config"]["layers"]
    flags = {l["name"]: l["config"].get("trainable", True) for l in cfg}
    assert flags["stage3_unit1_conv1"] is False and flags["bn_data"] is False
    assert flags["decoder_stage0a_conv"] is True and flags["final_conv"] is True
    assert keras_h5.load_keras_h5(path).encoder_freeze is True
    m.encoder_freeze = False
    keras_h5.save_keras_h5(m, path)
    assert keras_h5.load_keras_h5(path).encoder_freeze is False
