"""Native re-creation of the reference's missing `YOLOTensorRT` package (empty directory in the tree;
SURVEY.md F3).  Same import names as app.py:16-17 / test.py:5-6; no TensorRT: the detector runs on
hand-written gfx950 kernels."""
