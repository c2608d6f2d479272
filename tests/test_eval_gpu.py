"""Evaluation path on the MI355X: fused up-sample + argmax + confusion matrix, ConfMatrix.update, test()."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag,K", [("voc", 21), ("city", 19)])
def test_fused_confusion_matrix_matches_reference_fixture(tag, K):
    """Integer work: the matrix must be bit-exact against the reference's ConfMatrix (captured in eval.npz)."""
    from css_amd.util import ConfMatrix, mean_intersection_over_union
    g = np.load(os.path.join(GOLD, "eval.npz"))
    fused, plain = ConfMatrix(K, ":6.4f", "a"), ConfMatrix(K, ":6.4f", "b")
    for bi in range(2):
        pred = torch.from_numpy(g[f"{tag}::pred{bi}"]).to(dev())
        lab = torch.from_numpy(g[f"{tag}::lab{bi}"].astype(np.int64)).to(dev())
        # logical NCHW with channels_last memory, as the HIP model returns it
        pred_cl = pred.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        am = fused.update_from_logits(pred_cl, lab, return_argmax=True)
        assert torch.equal(am.cpu(), torch.from_numpy(g[f"{tag}::argmax{bi}"]))
        plain.update(am.flatten().long(), lab.flatten())
    want = torch.from_numpy(g[f"{tag}::mat"])
    assert torch.equal(fused.mat.cpu(), want)
    assert torch.equal(plain.mat.cpu(), want)
    assert abs(mean_intersection_over_union(fused.mat) - float(g[f"{tag}::miou"])) < 1e-6
    assert "a " in str(fused)


def test_confusion_matrix_edge_cases():
    from css_amd.util import ConfMatrix
    K = 5
    m = ConfMatrix(K)
    # all labels ignored -> zero matrix; labels == K and 255 are dropped like negative ones
    pred = torch.randn(1, K, 3, 3, device=dev())
    lab = torch.tensor([[[-1, 255, K], [-1, -1, -1], [K, K, 255]]], device=dev())
    m.update_from_logits(pred, lab)
    assert int(m.mat.sum()) == 0
    # 1x1 prediction map broadcast to the label size, bf16 logits
    pred = torch.zeros(2, K, 1, 1, device=dev(), dtype=torch.bfloat16)
    pred[0, 3] = 1
    pred[1, 1] = 1
    lab = torch.full((2, 4, 6), 2, device=dev())
    m.update_from_logits(pred, lab)
    assert int(m.mat[2, 3]) == 24 and int(m.mat[2, 1]) == 24 and int(m.mat.sum()) == 48
    # ties resolve to the first maximum, like torch.argmax
    pred = torch.zeros(1, K, 2, 2, device=dev())
    m2 = ConfMatrix(K)
    m2.update_from_logits(pred, torch.zeros(1, 2, 2, dtype=torch.int64, device=dev()))
    assert int(m2.mat[0, 0]) == 4
    # h == K (and w == K): the layout is taken from the argument, never guessed from the shape
    pred = torch.randn(1, K, K, K, device=dev())
    lab = pred.argmax(1)
    m3, m4 = ConfMatrix(K), ConfMatrix(K)
    m3.update_from_logits(pred, lab)
    m4.update_from_logits(pred.permute(0, 2, 3, 1).contiguous(), lab, channels_last=True)
    assert int(m3.mat.diag().sum()) == K * K and torch.equal(m3.mat, m4.mat)
    with pytest.raises(ValueError):
        ConfMatrix(K).update_from_logits(torch.zeros(1, 3, 3, K + 1, device=dev()), torch.zeros(1, 3, 3, dtype=torch.int64, device=dev()))
    with pytest.raises(Exception):
        ConfMatrix(K).update(torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.int64))   # CPU tensors: no CPU path


def test_eval_loop_equals_oracle_on_a_synthetic_loader():
    """css_amd.evaluate.test() (EMA model in eval mode, running-statistics BN) against the oracle's eval forward + confusion matrix."""
    from css_amd.evaluate import test as run_eval
    from css_amd.networks import resnet
    from css_amd.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep
    from oracle import css_oracle as O
    K, S = 21, 65
    sd = O.init_state("tv", K, 256, 9, 0.25)
    net = DeepLabv3Plus_with_rep(resnet.resnet101_tv(), dilate_scale=8, num_classes=K)
    net.load_state_dict(sd)
    net = net.to(dev()).train()
    g = torch.Generator().manual_seed(2)
    loader = [(torch.randn(2, 3, S, S, generator=g), torch.randint(-1, K, (2, S, S), generator=g)) for _ in range(2)]
    miou = run_eval(loader, net, {"Network": {"num_class": K}})
    assert net.training                                   # restored
    mat = torch.zeros(K, K, dtype=torch.int64)
    for x, y in loader:
        pred, _ = O.deeplab_forward(sd, x, "tv", False, K, 256)
        m, _ = O.eval_confusion(pred, y, K)
        mat += m
    ref = O.mean_iou(mat)
    # random-init logits are nearly flat across classes: a handful of arg-max decisions can differ at the 1e-5 level of fp32
    assert abs(miou - ref) < 2e-3, (miou, ref)
