"""Exception types raised on the hot path; names, hierarchy, messages and attributes follow
stainlib/utils/excepts.py:5-23 because user code catches them by name."""


class DigitalPathologyError(Exception):
    """Root of the library's error hierarchy."""


class DigitalPathologyAugmentationError(DigitalPathologyError):
    """Base class of augmentation errors."""


class InvalidRangeError(DigitalPathologyAugmentationError):
    """A sigma / bias / cutoff interval is malformed.  ``title`` names the interval, ``range`` is it."""

    def __init__(self, title, range):  # noqa: A002  (the reference's keyword name)
        self.title = title
        self.range = range
        DigitalPathologyAugmentationError.__init__(self, "Invalid range of {}: {}".format(title, range))


class TissueMaskException(Exception):
    """No pixel passed the luminosity threshold (stain_utils.py:46-47)."""
