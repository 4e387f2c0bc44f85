"""CPU oracle for the detect -> NMS/inflate/crop -> ViT-classify hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``yolov8-vit_amd/``)
may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Every function cites the reference file:line it restates (paths are relative
to the reference checkout).  Pinning status per function is stated in its
docstring:

* "pinned"   - checked against golden vectors captured from the reference's
               own Python functions (tests/golden/make_golden.py, run in the
               build container where the reference is mounted);
* "unpinned" - the arithmetic lives in a third-party package that is absent
               both from the reference tree and from this image (timm,
               ultralytics, torchvision, cv2, albumentations, TensorRT); the
               restatement follows the published algorithm and the reference's
               own call sites / prose, and parity is "unpinned".
"""
