"""mIoU curves of the miniature training run (tests/mini_train.py) for the record under profiles/: one line per (dtype, lr, seed) run.
Usage: python scripts/mini_train_curves.py OUT.json [epochs] [lr,lr,...] [dtypes: bf16,f32] [S]"""
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mini_train as M  # noqa: E402

if __name__ == "__main__":      # (guarded: MINI_MP_CONTEXT=spawn / forkserver re-import this file in the loader workers)
    out = sys.argv[1]
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    lrs = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0.02]
    dts = sys.argv[4].split(",") if len(sys.argv) > 4 else ["bf16"]
    S = int(sys.argv[5]) if len(sys.argv) > 5 else 129
    d = tempfile.mkdtemp()
    ds = M.make_dataset(d + "/voc", d + "/txt", S=S)
    res = []
    for dt in dts:
        for lr in lrs:
            t = time.time()
            r = M.run(ds, torch.bfloat16 if dt == "bf16" else torch.float32, epochs, lr=lr, log=lambda s: print(dt, lr, s, flush=True),
                      workers=int(os.environ.get("MINI_WORKERS", "0")), mp_context=os.environ.get("MINI_MP_CONTEXT") or None)
            r.pop("trainer")
            r.update(dtype=dt, lr=lr, seconds=time.time() - t, S=S, epochs=epochs)
            print(dt, lr, "curve", [round(x, 3) for x in r["curve"]], "seconds", round(r["seconds"], 1), "timing", r["timing"], flush=True)
            res.append(r)
            torch.cuda.empty_cache()
    json.dump(res, open(out, "w"))
