"""pnnp_amd -- MI355X-native implementation of the PNNP data-parallel hot path.

Host-side mirror of the reference's operator interface (same names and argument
meaning) over hand-written HIP kernels for gfx950, reached through the C ABI declared
in include/pnnp_hip.h.  No CPU fallback: every compute entry point raises if the HIP
library is missing or a tensor is not on the GPU.
"""
__version__ = '0.1.0'
