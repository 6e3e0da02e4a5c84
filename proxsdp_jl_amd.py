"""Import shim: the package directory is `proxsdp.jl_amd/` (the name the build
contract fixes), which is not a valid Python identifier.  `import
proxsdp_jl_amd` loads that directory as a regular package under this name."""
import importlib.util
import pathlib
import sys

_dir = pathlib.Path(__file__).resolve().parent / "proxsdp.jl_amd"
_spec = importlib.util.spec_from_file_location(
    __name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
