"""`det_postprocess` of the missing package (解读.md:25,82-84): slice the 4 engine outputs to num_dets."""
import torch


def det_postprocess(data):
    assert len(data) == 4
    num_dets, bboxes, scores, labels = (d[0] for d in data)
    n = int(num_dets.item())
    if n == 0:
        return bboxes.new_zeros((0, 4)), scores.new_zeros((0,)), labels.new_zeros((0,))
    return bboxes[:n], scores[:n], labels[:n]
