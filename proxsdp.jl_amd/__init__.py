"""proxsdp.jl_amd -- MI355X-native PDHG / PSD-projection engine behind ProxSDP's
solver boundary (`chambolle_pock`, /root/reference/src/MOI_wrapper.jl:310).

Only what the hot path needs lives here:
  csrc/        hand-written HIP kernels (gfx950) + the C-ABI library libproxsdp_hip.so
  binding.py   ctypes binding of include/proxsdp_hip.h (fails loudly without the .so)
  optimizer.py host-side mirror of the reference's `ProxSDP.Optimizer` surface
  problems.py  standard-form container + instance generators
"""
from . import problems  # noqa: F401
