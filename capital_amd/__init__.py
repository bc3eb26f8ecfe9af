"""capital_amd -- MI355X-native replacement for the BLAS/LAPACK/MPI layer under
huttered40/capital's recursive Cholesky (cholinv) and CA-CholeskyQR2 (cacqr).

The product is libcapital_hip.so (C-ABI in include/capital_hip.h, hand-written HIP for
gfx950) plus the C++ host headers in capital_amd/src that mirror the reference's call
surface.  This Python package is plumbing: it loads the library and hands it device
pointers.  There is no CPU fallback anywhere in this package.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
