"""Host-side logic and the C-ABI boundary, without a GPU: the library loads and
exports every symbol include/miseg.h declares, the ctypes descriptor matches
the C struct, and the utils.model surface behaves like the reference's
(/root/reference utils/model.py) for construction, reshaping, persistence and
errors."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    from miseg_amd import lib as L
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "miseg.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(miseg_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libmiseg.so does not export %s" % name
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    assert lib.miseg_version() == 100


def test_conv_desc_layout_matches_c(pkg, tmp_path):
    import ctypes
    from miseg_amd import lib as L
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "miseg.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(miseg_conv_desc),'
                   'offsetof(miseg_conv_desc,N), offsetof(miseg_conv_desc,in_img_stride),'
                   'offsetof(miseg_conv_desc,res_img_stride), offsetof(miseg_conv_desc,tile_hint));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = list(map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()))
    D = L.ConvDesc
    assert got == [ctypes.sizeof(D), D.N.offset, D.in_img_stride.offset, D.res_img_stride.offset,
                   D.tile_hint.offset]


def test_plan_records_and_validates_without_gpu(pkg):
    """Descriptor validation is host code: bad arguments are refused with a message."""
    import ctypes as C
    from miseg_amd import lib as L
    lib = L.load()
    plan = C.c_void_p(lib.miseg_plan_create())
    d = L.ConvDesc()
    assert lib.miseg_plan_add_conv(plan, C.byref(d)) == -1
    assert b"null" in lib.miseg_last_error()
    d.inp = d.weight = d.out = 4096
    d.N = d.H = d.W = d.Ho = d.Wo = 4
    d.Cout, d.Cin = 8, 24
    d.KH = d.KW = d.stride_y = d.stride_x = 1
    assert lib.miseg_plan_add_conv(plan, C.byref(d)) == -2          # Cin % 32
    d.Cin = 32
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 32, 128, 512
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = 8, 32, 128
    d.temperature = 1.0
    assert lib.miseg_plan_add_conv(plan, C.byref(d)) == 0
    assert lib.miseg_plan_num_ops(plan) == 1
    assert lib.miseg_plan_op_macs(plan, 0) == 4 * 4 * 4 * 8 * 32
    lib.miseg_plan_destroy(plan)


@pytest.mark.parametrize("enc,dec,total", [
    ("resnet18", "U-Net", 22903084), ("resnet34", "U-Net", 33018668),
    ("densenet169", "Linknet", 17051571), ("densenet169", "FPN", 15716035),
    # bottleneck ResNets: the two independently written builders (graph, oracle spec) must agree; the
    # totals are what they agree on, not published numbers
    ("resnet50", "U-Net", 52490028), ("resnet101", "Linknet", 50266940), ("resnet152", "FPN", 61648204),
    ("vgg16", "U-Net", 33476003), ("vgg19", "FPN", 22884547)])
def test_graph_param_counts_and_oracle_spec(pkg, enc, dec, total):
    from miseg_amd import netspec
    from oracle import nets
    g = netspec.build_graph(enc, dec, (64, 96, 3), 3)
    assert g.count_params() == total
    spec = nets.build_spec(enc, dec, 3)
    assert [(l, w, tuple(s)) for l, w, s, _ in g.weight_spec()] == \
        [(l, w, tuple(s)) for l, (_, ws) in spec.items() for w, s, _ in ws]
    assert [i for _, _, _, i in g.weight_spec()] == [i for _, (_, ws) in spec.items() for _, _, i in ws]


def test_vgg_encoder_parameter_counts(pkg):
    """keras_applications' documented totals without the classifier head: VGG16 14,714,688 and VGG19
    20,024,384 parameters."""
    from miseg_amd import netspec
    for enc, total in (("vgg16", 14714688), ("vgg19", 20024384)):
        g = netspec.build_graph(enc, "FPN", (64, 64, 3), 3)
        assert sum(g.layers[n].count_params() for n in g.encoder_layers) == total
        assert g.layers[g.encoder_output].cls == "MaxPooling2D"
        # segmentation_models takes Backbones.get_feature_layers(name, n=4): block1_conv2, the fifth layer
        # its table lists, is not a skip
        assert [g.layers[n].out.c for n in g.skip_layers] == [512, 512, 256, 128]
    # U-Net and Linknet put two centre blocks between a pooled encoder output and the decoder; the last
    # decoder stage has no skip (3x3x32x32 kernel in U-Net, 16 outputs and no Add in Linknet)
    g = netspec.build_graph("vgg16", "U-Net", (64, 64, 3), 3)
    assert "center_block2_relu" in g.layers and "decoder_stage4_concat" not in g.layers
    assert g.layers["decoder_stage4b_conv"].weights[0][1] == (3, 3, 32, 32)
    g = netspec.build_graph("vgg16", "Linknet", (64, 64, 3), 3)
    assert g.layers["decoder_stage4c_conv"].weights[0][1] == (1, 1, 32, 16) and "decoder_stage4_add" not in g.layers


def test_densenet_skips_resolve_by_keras_index(pkg):
    from miseg_amd import netspec
    g = netspec.build_graph("densenet169", "FPN", (64, 64, 3), 3)
    assert g.skip_layers == ["pool4_conv", "pool3_conv", "pool2_conv", "conv1/relu"]
    assert len(g.encoder_layers) == 595
    assert [g.layers[n].out.c for n in g.skip_layers] == [640, 256, 128, 64]


def test_make_model_surface(pkg, tmp_path):
    from utils.model import (CUSTOM_OBJECTS, METRICS, get_model_input_shape, load_model, load_models,
                             make_model, replace_model_input_shape, TemperatureScaledSoftmax)
    loss = CUSTOM_OBJECTS["categorical_crossentropy_plus_dice_loss"]
    m = make_model("resnet18", "U-Net", (64, 96, 3), 3, 1e-4, loss, METRICS, False, "AgNOR", seed=1)
    assert m.name == "AgNOR" and m.input_shape == (None, 64, 96, 3)
    assert get_model_input_shape(m) == (64, 96, 3)
    assert [x.name for x in m.metrics] == ["f1-score", "iou_score"]
    assert m.loss.name == "categorical_crossentropy_plus_dice_loss"
    assert m.optimizer.learning_rate == 1e-4 and m.optimizer.epsilon == 1e-7
    assert m.layers[-1].name == "softmax" and m.layers[-2].name == "final_conv"
    assert m.layers[-1].output.c == 3
    assert set(CUSTOM_OBJECTS) == {"categorical_crossentropy", "categorical_crossentropy_plus_dice_loss",
                                   "focal_loss_plus_dice_loss", "focal_loss", "f1-score", "iou_score",
                                   "TemperatureScaledSoftmax", "PapBias"}
    lines = []
    m.summary(print_fn=lines.append)
    assert any("Total params: 22903084" in s for s in lines)
    # he_uniform bound of the first decoder conv: sqrt(6 / fan_in)
    k = m.weights["decoder_stage0b_conv"]["kernel"]
    assert np.abs(k).max() <= np.sqrt(6.0 / (9 * 768)) and np.abs(k).max() > 0.9 * np.sqrt(6.0 / (9 * 768))
    # same weights at a new input size
    m2 = replace_model_input_shape(m, (128, 128, 3))
    assert m2.input_shape == (None, 128, 128, 3) and m2.name == "AgNOR"
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), m2.get_weights()))
    with pytest.raises(TypeError):
        replace_model_input_shape(m, None)
    # persistence
    path = tmp_path / "AgNOR_e001_l0.5000_vl0.6000.npz"
    m.save(path)
    m3 = load_model(str(path), input_shape=(32, 32, 3))
    assert m3.input_shape == (None, 32, 32, 3) and len(m3.metrics) == 2
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), m3.get_weights()))
    assert len(load_models(str(tmp_path))) == 1
    with pytest.raises(FileNotFoundError):
        load_model(str(tmp_path / "missing.npz"))
    with pytest.raises(RuntimeError):
        load_models(str(tmp_path / "missing_dir"))
    with pytest.raises(UnboundLocalError):
        make_model("resnet18", "PSPNet", (64, 64, 3), 3, 1e-4, loss, METRICS)
    # sigmoid for one class, temperature softmax on request
    assert make_model("resnet18", "Linknet", (64, 64, 3), 1, 1e-4, loss, []).layers[-1].name == "sigmoid"
    t = make_model("resnet18", "FPN", (64, 64, 3), 3, 1e-4, loss, [], use_tss=True)
    assert t.graph.temperature == 0.1 and t.layers[-2].name == "head_conv"
    assert TemperatureScaledSoftmax(0.1).get_config()["temperature"] == 0.1


def test_engine_refuses_to_run_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from miseg_amd import lib as L
    from utils.model import make_model
    m = make_model("resnet18", "U-Net", (64, 64, 3), 3, 1e-4, None, [])
    with pytest.raises(L.MisegError):
        m(np.zeros((1, 64, 64, 3), np.float32))
