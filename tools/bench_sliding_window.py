#!/usr/bin/env python3
"""BASELINE config 5: sliding-window inference on a synthetic full BraTS volume (1, 4, 240, 240, 155), overlap 0.5, sw_batch_size 2
(utils/inference_brats.py:209-216 with train_config batch_size 2) -> volumes/s, one JSON line.  argv: [roi=128|96] [overlap] [repeats]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from bench import WORKLOADS
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils import inference_runtime as IR
from veloxseg_amd.utils.metric.metrics_brats import cal_dice

roi = int(sys.argv[1]) if len(sys.argv) > 1 else 128
overlap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg, _ = WORKLOADS["brats128" if roi == 128 else "brats96"]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda().eval()
vol = torch.randn(1, 4, 240, 240, 155, device="cuda")
label = torch.randint(0, 4, (1, 1, 240, 240, 155), device="cuda")
starts = IR.window_starts((240, 240, 155), (roi,) * 3, IR.scan_interval((240, 240, 155), (roi,) * 3, overlap))
with torch.inference_mode():
    for _ in range(2):
        logits, labels = IR.infer_volume(model, vol, (roi,) * 3, 2, overlap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        logits, labels = IR.infer_volume(model, vol, (roi,) * 3, 2, overlap)
        dice = cal_dice(labels, label)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print(json.dumps({"metric": "sliding-window inference volumes/s (4x240x240x155, argmax + BraTS Dice included)", "value": round(1 / dt, 4), "unit": "volumes/s",
                  "ms_per_volume": round(dt * 1e3, 2), "windows": len(starts), "windows_per_s": round(len(starts) / dt, 2), "roi": roi, "overlap": overlap,
                  "sw_batch_size": 2, "dtype": "f32", "data": "synthetic randn volume, random-init weights", "dice_vs_random_labels": [round(v, 4) for v in dice]}))
