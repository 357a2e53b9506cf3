"""Import alias: the package lives in the directory ``zk-dcap-verifier_amd/`` (the name the build
contract fixes), which is not a valid Python identifier.  ``import zk_dcap_verifier_amd`` loads it."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "zk-dcap-verifier_amd")
_spec = _u.spec_from_file_location("zk_dcap_verifier_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["zk_dcap_verifier_amd"] = _mod
_spec.loader.exec_module(_mod)
