"""`TRTModule` stand-in (app.py:17,28-29; KAT-2 I/O contract test.ipynb:20-24).

`TRTModule(weight, device)` loads YOLOv8 weights instead of a serialized TensorRT engine:
  * a file written by `torch.save(state_dict)` in the ultralytics key layout (fused `model.N.conv.weight/bias`
    or unfused conv+bn: BatchNorm is folded on load, eps 1e-3), read with `weights_only=True`;
  * or a dict with the same keys; or the string "random[:scale[:nc[:seed]]]" for synthetic weights.
A pickled ultralytics `best.pt` (a whole model object) cannot be read without ultralytics and is refused.
"""
from collections import namedtuple
from typing import Dict, List, Optional, Union

import torch

import yvhip
from yvhip import engines

Tensor = namedtuple('Tensor', ('name', 'dtype', 'shape'))
_OUTPUTS = ['num_dets', 'bboxes', 'scores', 'labels']


def fold_batchnorm(sd: Dict[str, torch.Tensor], eps: float = 1e-3) -> Dict[str, torch.Tensor]:
    """conv(no bias)+BN -> conv with bias: w' = w*g/sqrt(v+eps), b' = beta - mu*g/sqrt(v+eps)."""
    out = dict(sd)
    for k in [k for k in sd if k.endswith(".bn.weight")]:
        p = k[:-len("bn.weight")]
        g, b = sd[p + "bn.weight"].float(), sd[p + "bn.bias"].float()
        mu, var = sd[p + "bn.running_mean"].float(), sd[p + "bn.running_var"].float()
        s = g / torch.sqrt(var + eps)
        out[p + "conv.weight"] = sd[p + "conv.weight"].float() * s.view(-1, 1, 1, 1)
        out[p + "conv.bias"] = b - mu * s
        for t in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out.pop(p + "bn." + t, None)
    return out


def _guess_scale_nc(sd):
    c0 = sd["model.0.conv.weight"].shape[0]
    scale = {16: "n", 32: "s", 48: "m"}.get(int(c0))
    if scale is None:
        raise yvhip.YvError(f"unsupported YOLOv8 width (stem has {c0} channels)")
    return scale, int(sd["model.22.cv3.0.2.weight"].shape[0])


class TRTModule(torch.nn.Module):
    def __init__(self, weight: Union[str, dict], device: Optional[torch.device] = None, size: int = 640) -> None:
        super().__init__()
        self.device = torch.device(device) if device is not None else torch.device('cuda:0')
        if isinstance(weight, dict):
            sd = weight
        elif isinstance(weight, str) and weight.startswith("random"):
            parts = weight.split(":")
            scale = parts[1] if len(parts) > 1 else "n"
            nc = int(parts[2]) if len(parts) > 2 else 5
            seed = int(parts[3]) if len(parts) > 3 else 42
            sd = engines.init_yolo_state(scale, nc, seed=seed, head_gain=4.0)
        else:
            sd = torch.load(str(weight), map_location="cpu", weights_only=True)
            if not isinstance(sd, dict) or "model.0.conv.weight" not in sd:
                raise yvhip.YvError(f"{weight}: expected a YOLOv8 state dict (ultralytics key layout)")
        sd = fold_batchnorm(sd)
        self.scale, self.nc = _guess_scale_nc(sd)
        self.engine = engines.YoloEngine(sd, self.scale, self.nc, size, device=str(self.device))
        self.size = size
        self.inp_info = [Tensor('images', torch.float32, (1, 3, size, size))]
        self.out_info = [Tensor('num_dets', torch.int32, (1, 1)), Tensor('bboxes', torch.float32, (1, 100, 4)),
                         Tensor('scores', torch.float32, (1, 100)), Tensor('labels', torch.int32, (1, 100))]
        self.idx = list(range(4))
        self.score_threshold, self.iou_threshold, self.topk = 0.25, 0.65, 100     # test.ipynb:1771-1773

    def set_desired(self, desired: Optional[List[str]]):
        if isinstance(desired, (list, tuple)) and len(desired) == 4:
            self.idx = [_OUTPUTS.index(n) for n in desired]

    def forward(self, *inputs):
        """(B,3,S,S) float in [0,1] (the `blob`) or (B,S,S,3) uint8 -> (num_dets, bboxes, scores, labels)."""
        x = inputs[0]
        if x.dtype != torch.uint8:                              # blob = u8/255 exactly, so this is lossless
            x = (x.to(self.device).float() * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
        x = x.to(self.device).contiguous()
        boxes, scores = self.engine(x)
        outs = yvhip.efficient_nms(boxes, scores, self.score_threshold, self.iou_threshold, self.topk)
        outs = [outs[i] for i in self.idx]
        return tuple(outs)
