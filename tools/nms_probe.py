"""How much work does the NMS stage see in the headline bench?  Candidate counts (score > 0.25) per image and per class,
detections kept, for the bench's synthetic detector (random-init YOLOv8n, head gain 4) and the bench's images."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
import torch  # noqa: E402
import yvhip  # noqa: E402
from yvhip import engines  # noqa: E402

dev = "cuda:0"
yolo = engines.YoloEngine(engines.init_yolo_state("n", 5, seed=42, head_gain=4.0), "n", 5, 640, dev)
g = torch.Generator().manual_seed(1234)
images = torch.randint(0, 256, (32, 640, 640, 3), generator=g, dtype=torch.uint8).to(dev)
boxes, scores = yolo(images)
torch.cuda.synchronize()
cand = (scores > 0.25)
per_img = cand.flatten(1).sum(1)
per_cls = cand.sum(1)
num, bb, sc, lb = yvhip.efficient_nms(boxes, scores)
print("candidates per image: min %d median %d max %d" % (int(per_img.min()), int(per_img.median()), int(per_img.max())))
print("candidates per (image, class): max %d ; per-class mean %s" % (int(per_cls.max()), per_cls.float().mean(0).tolist()))
print("num_dets:", num.flatten().tolist())
for b in range(2):
    n = int(num[b, 0])
    print("image", b, "kept labels histogram", torch.bincount(lb[b, :n].long(), minlength=5).tolist(),
          "score range", float(sc[b, :n].min()) if n else None, float(sc[b, :n].max()) if n else None)
