"""Import shim: exposes the package that lives in the hyphen-named directory
`deep-convolutional-neural-network-resnet-26-and-attention-network_amd/` as the module `mil_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "deep-convolutional-neural-network-resnet-26-and-attention-network_amd")
_spec = importlib.util.spec_from_file_location("mil_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mil_amd"] = _mod
_spec.loader.exec_module(_mod)
