"""Process-wide device contexts (one mp_ctx per GPU, created on first use)."""
import os

from . import native

_CONTEXTS = {}


def default_device():
    """LOCAL_RANK when launched by torch.distributed.run (one process per GPU), else 0."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def get_context(device=None):
    """The shared :class:`native.Context` of ``device``; raises if libmi355plan.so or the GPU is missing."""
    device = default_device() if device is None else int(device)
    ctx = _CONTEXTS.get(device)
    if ctx is None or getattr(ctx, "_h", None) is None:
        ctx = native.Context(device)
        _CONTEXTS[device] = ctx
    return ctx


def close_all():
    for ctx in _CONTEXTS.values():
        ctx.close()
    _CONTEXTS.clear()
