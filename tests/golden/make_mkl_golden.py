"""Regenerates tests/golden/capital_mkl_schedules.npz: the WHOLE schedules (recursive Cholesky with inverse, cholinv.hpp:6-183;
1-D CholeskyQR2, cacqr.hpp:7-29,174-193) executed with the reference's actual arithmetic provider -- the cblas_* / LAPACKE_*
entry points of Intel MKL's libmkl_rt.so (its un-vendored third-party dependency, src/util/shared.h:24) -- bound at run time
underneath the schedule restatement of oracle/capital_oracle.c (orc_host_blas_bind).  Inputs are the reference's generators.

This is the strongest pin the image allows: the reference itself cannot be compiled here (no mkl.h, and stand-in headers are
not allowed), so these are NOT outputs of the reference binary; they are its schedule on its BLAS.  tests/test_golden.py
checks the oracle's own kernels (CPU) and the HIP path (MI355X) against them to 1e-12.

    python tests/golden/make_mkl_golden.py        (needs /opt/conda/lib/libmkl_rt.so.1; prints the library's version string)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402

O.build()
lib = O.bind_host_blas(1)          # one thread: MKL's summation order then does not depend on the machine's core count
assert lib is not None and "Math Kernel Library" in lib["library"], f"MKL not found (got {lib})"
print("host library:", lib["library"], lib["path"])
out = {"library": np.array(lib["library"])}
# (n, complete_inv, split, bc_mult_dim): one case without recursion, recursive ones with both inverse settings and a split of 2
for n, ci, split, bc in ((160, 0, 1, -2), (192, 1, 1, -3), (130, 1, 2, -2), (96, 1, 1, 0)):
    A = O.distribute_symmetric(n, n, 0, 0, 1, 1)
    R, Ri, info = O.cholinv_factor(A, ci, split, bc, 1, 1)
    assert info == 0 and O.cholesky_residual(A, R) <= 1e-15
    key = f"chol_n{n}_ci{ci}_s{split}_bc{-bc}"
    iu = np.triu_indices(n)                                   # both factors are upper triangular: the packed triangles are kept
    assert not np.tril(R, -1).any() and not np.tril(Ri, -1).any()
    out[key + "_R"], out[key + "_Rinv"] = R[iu], Ri[iu]
for m, n, variant in ((1536, 48, 2), (1000, 24, 1)):
    A = O.distribute_random(n, m, 0, 0, 1, 1, key=0)
    Q, R, info = O.cacqr_factor_1d(A, 1, variant)
    assert info == 0
    key = f"cqr_m{m}_n{n}_v{variant}"
    out[key + "_Q"], out[key + "_R"] = Q, R[np.triu_indices(n)]
np.savez_compressed(os.path.join(HERE, "capital_mkl_schedules.npz"), **out)
print("wrote", os.path.join(HERE, "capital_mkl_schedules.npz"), {k: v.shape for k, v in out.items() if v.ndim})
