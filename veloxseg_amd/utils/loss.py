"""Deep-supervised CE + Dice, MSE reconstruction and SDKT Gram loss (reference: utils/loss.py:10-86), one fused
HIP autograd op (veloxseg_amd.functional.veloxseg_loss)."""
import torch
from torch import nn

from .. import functional as VF
from .runtime import normalized_deep_loss_weights, veloxseg_output_layout


class Loss(nn.Module):
    """Same constructor / call signature as the reference: Loss(args, config, device, num_modal)(output, labels, sr_labels)."""

    def __init__(self, args, config, device=None, num_modal=2):
        super().__init__()
        self.model_name = getattr(args, "model_name", "VeloxSeg")
        if self.model_name != "VeloxSeg":
            raise NotImplementedError("veloxseg_amd.Loss implements the VeloxSeg branch of utils/loss.py only")
        self.device = device
        self.num_modal = num_modal
        self.register_buffer("deep_loss_weight", torch.tensor(config["deep_Loss_weight"], dtype=torch.float32))
        self._deep_w = [float(w) for w in config["deep_Loss_weight"]]
        self.rc_loss_weight = config.get("RC_Loss_weight")
        self.feature_loss_weight = config.get("Feature_Loss_weight")

    def cal_loss(self, output, labels, sr_labels=None):
        layout = veloxseg_output_layout(len(output), self.num_modal)
        a, b = layout["seg"]
        w = normalized_deep_loss_weights(self._deep_w, b - a)
        return VF.veloxseg_loss(list(output), labels, sr_labels, w, self.rc_loss_weight, self.feature_loss_weight, self.num_modal)

    def staged(self, n_heads):
        """the same loss as separable pieces for the staged training step (functional.StagedLoss): head weights, w_rc, w_f, M"""
        return VF.StagedLoss(normalized_deep_loss_weights(self._deep_w, n_heads), self.rc_loss_weight, self.feature_loss_weight, self.num_modal)

    def forward(self, output, labels, sr_labels=None):
        return self.cal_loss(output, labels, sr_labels)
