"""ckb_zkp_amd — MI355X-native (gfx950) MSM + NTT proving backend for sec-bit/ckb-zkp's Groth16/Marlin hot path.

The compute path is the hand-written HIP library `lib/libzkp_accel.so` behind the C ABI in
`include/zkp_accel.h`; this package is the host-side mirror of the reference's prover interface plus ctypes
plumbing.  There is no CPU fallback: importing the API without the built library, or creating a `Context`
without a gfx950 device, raises.
"""
from .params import BN254, BLS12_381, get_curve  # noqa: F401

__all__ = ["BN254", "BLS12_381", "get_curve"]

import os as _os

# The prover keeps ~20 HIP streams busy; ROCm maps them onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams
# sharing a queue serialise.  Must be set before the HIP runtime initialises (i.e. before the first torch.cuda / HIP call).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
