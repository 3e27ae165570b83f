"""Stub: the golden generator never loads files through the reference."""


def imread(*a, **k):
    raise RuntimeError("imageio is not available in this container")


imwrite = imread
