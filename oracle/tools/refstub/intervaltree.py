"""Stub `intervaltree` (container-only test tooling, see pysam.py).  Half-open [begin, end)."""


class Interval:
    def __init__(self, begin, end, data=None):
        self.begin, self.end, self.data = begin, end, data


class IntervalTree:
    def __init__(self, intervals=()):
        self.iv = list(intervals)

    def overlaps(self, pos):
        return any(i.begin <= pos < i.end for i in self.iv)
