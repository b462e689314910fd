"""Import shim: the package directory is named `graphicalmodellearning.jl_amd` (it carries the
reference's name), which is not a valid Python identifier, so it is loaded here under the module
name `gml_amd`:   import gml_amd as gml;  gml.learn(samples, gml.RISE(), gml.HIP())"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graphicalmodellearning.jl_amd")
_spec = importlib.util.spec_from_file_location("gml_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gml_amd"] = _mod
_spec.loader.exec_module(_mod)
