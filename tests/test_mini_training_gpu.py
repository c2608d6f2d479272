"""An end-to-end miniature training run that reads an mIoU (the closable proxy of north_star's "mIoU within +-0.3 of the
reference" clause, whose real form needs Pascal VOC and ImageNet weights this pool does not have).

tests/mini_train.py runs the reference's main loop (mix_label.py:86-147) on a synthetic but learnable segmentation task, through every (f) row of
SURVEY 8 at once: VOC-shaped tree -> VOC_BuildData -> DataLoaders -> MixTrainer.step (device_aug='pil', cutmix) -> evaluate.test on the
EMA model -> save_checkpoint at the best mIoU -> load_checkpoint and continue.  129 x 129 crops, B = 4 + 4, 6 epochs of 32 steps, lr 0.01
(loaders in the main process: see tests/mini_train.py on forked workers next to a live HIP context).

Asserted:
  (i)   the validation mIoU of the EMA model rises from near chance level (< 0.60 after the first epoch of 32 steps - measured 0.19 ... 0.36 over the builds of the round; 6 classes) to >= 0.80 - in fp32 AND in bf16;
  (ii)  |mIoU_bf16 - mIoU_fp32| is within max(0.03, 3 x d), d = |mIoU of two fp32 runs whose input batches differ by a 1-ulp relative perturbation|
        - the chaos floor of the problem itself (two correct fp32 runs), measured in the same test from the same seeds;
  (iii) a run resumed from the epoch-3 checkpoint (exact-resume format, css_amd/checkpoint.py) reproduces the un-interrupted run BIT FOR BIT:
        student, teacher, momentum, prototypes, batch-norm buffers and the remaining mIoU curve;
  (iv)  best_model.pth is in the reference's wire format and evaluating its 'ema_model' reproduces the best mIoU exactly.
Curves of one run of this test: profiles/r06_mini_training_curves.json.
"""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mini_train as M  # noqa: E402

EPOCHS, LR = 6, 0.01


def _strip(r):
    r = dict(r)
    r.pop("trainer", None)
    return r


@pytest.mark.timeout(600)      # (the whole test takes ~60 s; a loader that hands batches over in unpinned shared memory once stalled a copy for 20 minutes)
def test_mini_training_reads_an_miou_in_fp32_and_bf16_and_resumes_bit_for_bit(tmp_path):
    from css_amd import evaluate
    ds = M.make_dataset(str(tmp_path / "voc"), str(tmp_path / "txt"))
    ckdir = str(tmp_path / "ck")
    os.makedirs(ckdir)
    log = lambda tag: (lambda s: print(tag, s, flush=True))

    a = M.run(ds, torch.bfloat16, EPOCHS, lr=LR, ckpt_dir=ckdir, snapshot_epoch=2, log=log("bf16        "))
    # (iv) the best checkpoint, in the reference's format, holds the teacher that scored the best mIoU
    sd = torch.load(os.path.join(ckdir, "best_model.pth"), map_location="cpu", weights_only=False)
    assert set(sd) == {"epoch", "model", "ema_model", "optimizer", "lr_scheduler", "prototypes"}
    tr = a.pop("trainer")
    tr.model.ema_model.load_state_dict(sd["ema_model"])
    tr.model.refresh_weights()
    cfg = {"Network": {"num_class": M.K}}
    from css_amd.dataset_helpers import VOC
    test_set = VOC.VOC_BuildData(**ds).build()[2]
    again = float(evaluate.test(torch.utils.data.DataLoader(test_set, batch_size=4, drop_last=True, num_workers=0), tr.model.ema_model, cfg))
    assert again == a["best"], (again, a["best"])
    assert sd["epoch"] == a["curve"].index(a["best"]) + 1
    del tr, sd
    torch.cuda.empty_cache()

    # (iii) resume at epoch 3 from the snapshot and finish: bit-identical to the run that never stopped
    r = _strip(M.run(ds, torch.bfloat16, EPOCHS, lr=LR, resume=os.path.join(ckdir, "snap.pth"), log=log("bf16 resumed")))
    assert r["curve"] == a["curve"][3:], (r["curve"], a["curve"])
    for key in ("fp_student", "fp_teacher", "fp_momentum", "fp_proto", "fp_bn", "it"):
        assert r[key] == a[key], key
    torch.cuda.empty_cache()

    f = _strip(M.run(ds, torch.float32, EPOCHS, lr=LR, log=log("fp32        ")))
    torch.cuda.empty_cache()
    g = _strip(M.run(ds, torch.float32, EPOCHS, lr=LR, perturb=123, log=log("fp32 + 1 ulp")))
    torch.cuda.empty_cache()

    # (i) it learns, in both dtypes
    for name, run in (("bf16", a), ("fp32", f), ("fp32 + 1 ulp", g)):
        print(name, "mIoU per epoch:", [round(x, 4) for x in run["curve"]])
        assert run["curve"][0] < 0.60 and run["best"] >= 0.80 and run["curve"][-1] >= 0.75, (name, run["curve"])
    # (ii) bf16 sits inside the problem's own fp32 noise
    d_floor = max(abs(f["curve"][-1] - g["curve"][-1]), abs(f["best"] - g["best"]))
    bound = max(0.03, 3 * d_floor)
    d_final, d_best = abs(a["curve"][-1] - f["curve"][-1]), abs(a["best"] - f["best"])
    print(f"final mIoU: bf16 {a['curve'][-1]:.4f} fp32 {f['curve'][-1]:.4f} fp32+1ulp {g['curve'][-1]:.4f}; best: {a['best']:.4f} {f['best']:.4f} {g['best']:.4f}; "
          f"fp32 floor {d_floor:.4f}, bound {bound:.4f}, bf16 - fp32: final {d_final:.4f} best {d_best:.4f}")
    assert d_final <= bound and d_best <= bound, (d_final, d_best, bound)
    out = os.environ.get("CSS_MINI_TRAIN_OUT")
    if out:
        json.dump(dict(epochs=EPOCHS, lr=LR, steps_per_epoch=32, crop=129, batch="4+4", bf16=_strip(a), bf16_resumed_from_epoch_3=r, fp32=f, fp32_plus_1ulp=g,
                       fp32_floor=d_floor, bound=bound), open(out, "w"), indent=1)
