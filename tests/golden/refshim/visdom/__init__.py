"""Stub of the `visdom` package (not installed here): /root/reference/main.py imports it at the top but only
touches it under --plot, which the golden generators never pass."""


class Visdom(object):
    def __init__(self, *a, **kw):
        raise RuntimeError("visdom stub: --plot is not available")
