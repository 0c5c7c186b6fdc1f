#!/usr/bin/env python3
"""Inference entrypoint with the reference's command line (run_test.py): sliding-window segmentation + Dice.

The NIfTI readers of utils/inference_brats.py / inference_petct.py are out of scope; volumes are synthetic
(--synthetic N volumes of --volume_shape) unless --data_module names a module providing `iter_volumes(args, train_config, test_config)`
yielding (name, inputs (1, C, D, H, W), label (1, 1, D, H, W)).  The rest follows utils/inference_brats.py:190-255: eval mode,
sliding_window_predict(inputs, net, patch_size, batch_size, test_config), argmax, per-volume Dice, CSV of the results.
"""
import argparse
import csv
import importlib
import json
import os
import time

SUPPORTED_DATASETS = ("AutoPETII", "Hecktor2022", "BraTS2021")

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset_name", type=str, required=True, choices=SUPPORTED_DATASETS)
    parser.add_argument("--model_name", type=str, required=True)
    parser.add_argument("--train_date", type=str, default=None)
    parser.add_argument("--model_index", type=str, default=None)
    parser.add_argument("--checkpoint_dir", type=str, default=None, help="directory holding <checkpoint_index>.pth")
    parser.add_argument("--checkpoint_index", type=str, default="val_best")
    parser.add_argument("--model_config", type=str, required=True)
    parser.add_argument("--train_config", type=str, required=True)
    parser.add_argument("--test_config", type=str, required=True)
    parser.add_argument("--gpu_id", type=str, default="0")
    parser.add_argument("--num_workers", type=int, default=8)
    parser.add_argument("--specific_sample", type=int, default=None)
    parser.add_argument("--select_modal", type=int, default=None)
    parser.add_argument("--use_hd95", type=int, default=None, help="HD95 needs medpy (not available): must stay 0 / unset")
    parser.add_argument("--synthetic", type=int, default=2, help="number of synthetic volumes")
    parser.add_argument("--volume_shape", type=int, nargs=3, default=[240, 240, 155])
    parser.add_argument("--data_module", type=str, default=None)
    parser.add_argument("--out_csv", type=str, default=None)
    args = parser.parse_args()
    if args.use_hd95:
        raise SystemExit("HD95 (medpy) is out of scope of veloxseg_amd")
    cfgs = {}
    for k in ("model_config", "train_config", "test_config"):
        with open(getattr(args, k), "r", encoding="utf-8") as f:
            cfgs[k] = json.load(f)
    import veloxseg_amd  # noqa: F401
    import torch
    from veloxseg_amd.utils.inference_runtime import Net, sliding_window_predict
    from veloxseg_amd.utils.load_model import load_checkpoint, load_model
    torch.cuda.set_device(int(args.gpu_id.split(",")[0]))
    torch.manual_seed(12345)
    model = load_model(args.model_name, cfgs["model_config"])
    if args.checkpoint_dir:
        model = load_checkpoint(model, os.path.join(args.checkpoint_dir, args.checkpoint_index + ".pth"))
    net = Net(model.cuda().eval())
    mcfg = cfgs["model_config"][args.model_name]
    brats = args.dataset_name == "BraTS2021"
    if brats:
        from veloxseg_amd.utils.metric.metrics_brats import cal_dice
    else:
        from veloxseg_amd.utils.metric.metrics import metrics_tensor
    if args.data_module:
        volumes = importlib.import_module(args.data_module).iter_volumes(args, cfgs["train_config"], cfgs["test_config"])
    else:
        def volumes():
            g = torch.Generator().manual_seed(12345)
            for i in range(args.synthetic):
                x = torch.randn((1, sum(mcfg["in_ch"]), *args.volume_shape), generator=g)
                y = torch.randint(0, mcfg["n_classes"], (1, 1, *args.volume_shape), generator=g)
                yield f"synthetic_{i}", x, y
        volumes = volumes()
    patch = cfgs["train_config"]["patch_size"][args.dataset_name] if isinstance(cfgs["train_config"].get("patch_size"), dict) else mcfg["input_size"]
    rows = []
    with torch.inference_mode():
        for name, x, y in volumes:
            t0 = time.perf_counter()
            out = sliding_window_predict(x.cuda().float(), net, patch, cfgs["train_config"]["batch_size"], cfgs["test_config"])
            out = out.argmax(dim=1, keepdim=True)
            if brats:
                avg, et, tc, wt = cal_dice(out, y.cuda())
                rows.append([name, avg, et, tc, wt, float((out > 0).sum()), float((y > 0).sum())])
            else:
                fp, fn, _, _, _, iou, dice = metrics_tensor(y.cuda(), out)
                rows.append([name, dice, iou, fp, fn, float((out > 0).sum()), float((y > 0).sum())])
            torch.cuda.synchronize()
            print(name, [round(v, 4) if isinstance(v, float) else v for v in rows[-1][1:]], f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
    if args.out_csv:
        with open(args.out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["name", "Avg_Dice", "ET_Dice", "TC_Dice", "WT_Dice", "Prediction", "Label"] if brats else ["name", "Dice", "IoU", "FP", "FN", "Prediction", "Label"])
            w.writerows(rows)
