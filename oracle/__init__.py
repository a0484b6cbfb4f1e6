"""CPU oracle for the L1->L2 detector-calibration hot path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the arithmetic the reference
(Roman-HLIS-Cosmology-PIT/romanimpreprocess) performs on the path
``calibrateimage`` (``src/romanimpreprocess/L1_to_L2/gen_cal_image.py:480``):
reference-pixel correction, bias, Legendre linearity, IPC deconvolution,
fixed-weight ramp fit with jump detection, dark-rate subtraction and flat.
Every function cites the reference file:line it follows and spells out the
dtype of every intermediate (numpy-2 / NEP-50 promotion), because the HIP
kernels are held to the same per-operation rounding.

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- always as the checker or the reported
CPU baseline, never as the product.  ``romanimpreprocess_amd`` never imports it.

Parity status: PINNED.  ``tools/make_goldens.py`` imports the reference's own
``utils/fitting.py``, ``utils/ipc_linearity.py``, ``utils/flatutils.py`` and
``utils/reference_subtraction.py`` from ``/root/reference`` (with in-memory
stand-ins for the two I/O-only imports ``asdf`` and
``roman_datamodels.dqflags``), runs them on seeded inputs and commits inputs'
seeds + outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` holds
this package bit-identical to those outputs, plus the reference tests' own
literals (``tests/romanimpreprocess/test_linutils.py:14-48``).
Steps whose source is NOT in the reference tree (romancal dq-init, stcal
saturation flagging, romancal dark-rate subtraction / image-model packaging)
are restated from their call sites and documented behaviour: parity UNPINNED
for those.
"""

from . import finish, ipc, linearity, rampfit, refpix  # noqa: F401
from .chain import calibrate_arrays  # noqa: F401
