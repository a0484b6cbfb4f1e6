"""MI355X-native L1->L2 detector calibration for Roman WFI ramps (host side).

Drop-in for the per-pixel path of ``romanimpreprocess`` (``gen_cal_image.calibrateimage`` and the
``utils.fitting`` / ``utils.ipc_linearity`` / ``utils.reference_subtraction`` / ``utils.flatutils``
functions): the arithmetic runs in hand-written HIP kernels (gfx950) reached through the C-ABI
of ``libromanhip.so`` (``include/romanhip.h``).  There is no CPU fallback: importing the
numerics without the built library raises.
"""

__version__ = "0.1.0"
