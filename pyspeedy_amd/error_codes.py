"""`pyspeedy.error_codes` (pyspeedy/error_codes.py): the messages of the model's error codes (error_codes.f90:7-9), plus the one
code this library adds for a device model whose step could not be issued or checked at all (-3, include/pyspeedy_amd_driver.h)."""
from collections import defaultdict

from .speedy_driver import ERROR_CODES as _KNOWN


class _Codes(defaultdict):
    def __missing__(self, key):  # (the reference's defaultdict is given a one-argument factory and raises instead of answering)
        return "Unexpected error: %r" % (key,)


ERROR_CODES = _Codes(None, _KNOWN)
