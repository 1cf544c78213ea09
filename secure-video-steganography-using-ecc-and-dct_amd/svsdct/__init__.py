"""svsdct - host side of the MI355X block-DCT / QIM frame operator.

`native`  ctypes binding of libsvsdct.so (include/svsdct.h); raises if the library is missing.
`batch`   array-level embed / extract over stacks of gray frames (host or device resident).
`synth`   synthetic frames / payload bits (NumPy twin of the on-device generators).
"""
__all__ = ["native", "batch", "synth"]
