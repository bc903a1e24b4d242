"""ecfft_amd — MI355X-native EXTEND / ENTER / EXIT of the Elliptic-Curve FFT (drop-in for that path of
andrewmilson/ecfft).  The product is the C-ABI library (include/ecfft_hip.h, ecfft_amd/csrc/);
this package is the thin host-side mirror of the crate's `FFTree` / `FftreeField` API used by tests
and bench.py."""
from .fftree import (FFTree, Field, Moiety, EcfftError, FIELDS, secp256k1, m31, lib, device_info,  # noqa: F401
                     TBL_F, TBL_RECOMBINE, TBL_DECOMPOSE, TBL_XNN_S, TBL_XNN_S_INV, TBL_Z0_S1, TBL_Z1_S0, TBL_Z0_INV_S1, TBL_Z1_INV_S0, TBL_Z0Z0, TBL_Z1Z1)
from . import build  # noqa: F401
