"""Where the HOST time of a miniature training step goes (tests/mini_train.py at 129^2, B = 4 + 4: the GPU needs ~30 ms, the step took 190):
cProfile over 16 steps after one warm epoch, the in-step augmentation 'pil' vs 'identity', and evaluate.test timed alone."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mini_train as M  # noqa: E402
from css_amd import evaluate  # noqa: E402
from css_amd.dataset_helpers import VOC  # noqa: E402
from css_amd.networks import resnet  # noqa: E402
from css_amd.networks.ddp_model import Model_mix  # noqa: E402
from css_amd.train_step import MixTrainer  # noqa: E402

dev = torch.device("cuda:0")
d = tempfile.mkdtemp()
ds = M.make_dataset(d + "/voc", d + "/txt")
train_l, train_u, test_set = VOC.VOC_BuildData(**ds).build()
for aug in ("pil", "identity"):
    cfg = {"Dataset": {"crop_size": [129, 129], "scale_size": [0.5, 1.5], "mix_mode": "cutmix", "device_aug": aug}, "Network": {"num_class": M.K}}
    M.seed_all(1)
    m = Model_mix(resnet.resnet101_tv(), num_classes=M.K, output_dim=256, ema_alpha=0.95, config=cfg, temp=0.25).to(dev)
    m.model.train(); m.ema_model.train()
    m.set_compute_dtype(torch.bfloat16)
    tr = MixTrainer(m, num_classes=M.K, lr=0.01, total_iter=1000, num_queries=256, num_negatives=512)
    ld = torch.utils.data.DataLoader(train_l, batch_size=4, drop_last=True, num_workers=0, shuffle=False)
    batches = []
    for i, (x, y) in enumerate(ld):
        batches.append((x.to(dev), y.to(dev)))
        if i == 3:
            break
    for i in range(8):
        tr.step(batches[i % 4][0], batches[i % 4][1], batches[(i + 1) % 4][0])
    torch.cuda.synchronize()
    t = time.time()
    for i in range(16):
        tr.step(batches[i % 4][0], batches[i % 4][1], batches[(i + 1) % 4][0])
    t_host = time.time() - t
    torch.cuda.synchronize()
    t_all = time.time() - t
    print(f"aug={aug}: 16 steps: host enqueue {t_host / 16 * 1e3:.1f} ms per step, with the final sync {t_all / 16 * 1e3:.1f} ms per step", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(16):
        tr.step(batches[i % 4][0], batches[i % 4][1], batches[(i + 1) % 4][0])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(18)
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.time()
        miou = evaluate.test(torch.utils.data.DataLoader(test_set, batch_size=4, drop_last=True, num_workers=0), m.ema_model, cfg)
        torch.cuda.synchronize()
        print(f"evaluate.test (16 images, 4 batches, no workers) call {rep}: {time.time() - t:.2f} s, mIoU {float(miou):.4f}", flush=True)
    del tr, m
    torch.cuda.empty_cache()
