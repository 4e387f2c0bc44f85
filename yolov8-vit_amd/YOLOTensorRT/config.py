"""Class table of the deployed detector/classifier (YOLOTensorRT_yolodet_py_解读.md:24; 小白项目指南.md:167-174)."""
CLASSES = ['good', 'broke', 'lose', 'uncovered', 'circle']
COLORS = [(0, 200, 0), (0, 0, 230), (230, 120, 0), (200, 0, 200), (0, 200, 230)]
