"""TEST INFRASTRUCTURE ONLY.  qtorch (unpinned in the reference's requirements) is absent offline; the one function the
reference's converter uses from it is restated here from its published behaviour: `float_quantize(x, exp, man, rounding="nearest")`
rounds every element to the nearest value (ties to even) of a low-precision float format with `exp` exponent bits and `man`
mantissa bits.  The converter only calls it with (4, 3) on values already clipped to ±448 (converter.py:322-324), where that grid
is the e4m3fn grid, so the rounding is delegated to torch's own e4m3fn conversion."""
import torch


def float_quantize(x, exp, man, rounding="nearest"):
    if (exp, man, rounding) != (4, 3, "nearest"):
        raise NotImplementedError("shim covers the converter's only use: float_quantize(x, 4, 3, rounding='nearest')")
    return x.to(torch.float8_e4m3fn).to(x.dtype)
