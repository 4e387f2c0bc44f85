"""Serialisation of engine steps at the Python boundary.

The reference calls its models from Flask request threads and from background threads without any lock
(app.py:50-61,99-100); torch modules tolerate that for inference.  The HIP engines replay a fixed launch list over
per-engine activation buffers, so two interleaved replays would mix their activations.  `StepGuard` makes one replay
atomic: a re-entrant lock held while the step is ENQUEUED, plus stream ordering for the case that consecutive entrants
enqueue on different HIP streams (the newcomer's stream waits for everything the previous holder enqueued).  Same-stream
entrants - the normal case, all threads on the default stream - pay one lock acquisition and nothing on the device.
"""
from __future__ import annotations

import threading

import torch


def _current_stream():
    return torch.cuda.current_stream() if torch.cuda.is_available() else None


class StepGuard:
    def __init__(self, stream_fn=_current_stream):
        self._lock = threading.RLock()
        self._depth = 0
        self._stream = None
        self._stream_fn = stream_fn
        self._owner = None
        self.entries = 0

    @property
    def held(self) -> bool:
        """True when the CALLING thread is inside the guard (its engine-owned results stay valid until it leaves)."""
        return self._owner == threading.get_ident() and self._depth > 0

    def __enter__(self):
        self._lock.acquire()
        self._depth += 1
        self._owner = threading.get_ident()
        if self._depth == 1:
            self.entries += 1
            cur = self._stream_fn()
            if cur is not None and self._stream is not None and self._stream != cur:
                cur.wait_stream(self._stream)         # buffers are shared: order this step behind the previous holder's
        return self

    def __exit__(self, *exc):
        if self._depth == 1:
            self._stream = self._stream_fn()
            self._owner = None
        self._depth -= 1
        self._lock.release()
        return False
