"""Test-only pass-through for `numba.jit` (numba is not installed in this image)."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f
    return deco
