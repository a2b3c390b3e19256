"""oracle/ -- TEST INFRASTRUCTURE ONLY.

ctypes access to (a) liboracle.so, the plain-C CPU restatement of the reference's
EM hot path, and (b) when present, oracle/_ref/libstrawberry_ref.so, the
reference's own code compiled from /root/reference by oracle/Makefile.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package.  The product (strawberry_amd/, libsbgpu.so) never does.
"""
from .lib import (  # noqa: F401
    OracleLib,
    RefLib,
    build,
    have_ref,
    SBO_EM_OK,
    SBO_EM_INIT_EMPTY,
    SBO_EM_DENOM_ZERO,
    SBO_EM_MAXITER,
)
