"""Binary segmentation metrics -- mirrors utils/metric/metrics.py of the reference (names, argument order, return order).

`metrics_tensor(gt, pred)` = [FP rate, FN rate, precision, recall, F1, IoU, Dice], each the mean over the batch of the per-sample value
with the reference's smoothing (1e-5) and float32 arithmetic (utils/metric/metrics.py:44-91).  The per-sample counts come from one
GPU pass (vx_confusion); the reference instead moves both volumes to the host as IntTensors.  HD95 (medpy) is out of scope.
"""
import torch

from ._counts import confusion, to_label_map


def metrics_tensor(gt: torch.Tensor, pred: torch.Tensor):
    assert len(gt.shape) == len(pred.shape)
    if pred.shape[1] == 2:
        pred = pred[:, 1:]
    if gt.shape[1] == 2:
        gt = gt[:, 1:]
    pred_l = pred.to(torch.int32) if pred.is_floating_point() else pred      # the reference truncates with .type(IntTensor)
    gt_l = gt.to(torch.int32) if gt.is_floating_point() else gt
    conf = confusion(pred_l, gt_l, 2)                                        # labels are {0, 1} on this path (binary datasets)
    tn, fp, fn, tp = conf[:, 0, 0], conf[:, 0, 1], conf[:, 1, 0], conf[:, 1, 1]
    gt_sum, pred_sum = tp + fn, tp + fp
    inter, union = tp, tp + fp + fn
    smooth = 1e-5
    precision = tp / (pred_sum + smooth)
    recall = tp / (gt_sum + smooth)
    f1 = 2 * precision * recall / (precision + recall + smooth)
    fpr = fp / (fp + tn + smooth)
    fnr = fn / (fn + tp + smooth)
    jaccard = inter / (union + smooth)
    dice = 2 * inter / (gt_sum + pred_sum + smooth)
    return [float(m.mean()) for m in [fpr, fnr, precision, recall, f1, jaccard, dice]]


def show_deep_metrics(outputs, labels, deep=True):
    """utils/metric/metrics.py:6-27"""
    if not isinstance(outputs, (list, tuple)):
        outputs = [outputs]
    res, string = None, ""
    for k, o in enumerate(outputs if deep else outputs[:1]):
        out = to_label_map(o)
        fp, fn, _, _, _, iou, dice = metrics_tensor(labels, out)
        string += f"[FP:{fp:.4f}, FN:{fn:.4f}, IoU:{iou:.4f}, Dice:{dice:.4f} pix:{int(out.sum()):6}/{int(labels.sum()):6}]\n"
        if k == 0:
            res = [fp, fn, iou, dice]
    return res, string + "\n"
