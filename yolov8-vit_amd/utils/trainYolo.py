"""Drop-in call surface of the reference's `utils.trainYolo` (utils/trainYolo.py:6-35,124-137).

The arithmetic of `train()` lives entirely in `ultralytics` (detection loss, assigner, augmentation,
EMA, AMP - SURVEY.md 8(f) N2, parity unpinned); this round builds the detector FORWARD path only, so the
training entry points keep their signatures and report that clearly instead of silently doing nothing.
"""
import yvhip


def train(epochs, batch, data):
    raise yvhip.YvError("YOLOv8 training (conv backward + v8 detection loss) is not built in this round")


def yoloRetrain():
    """app.py:99-100 runs this on a background thread and ignores the result."""
    try:
        train(epochs=1, batch=1, data="/app/train/new/data.yaml")
    except Exception as e:
        print(f"yoloRetrain: {e}")
        return False
    return True
