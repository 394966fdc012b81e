"""rwkv.cpp_amd -- MI355X-native drop-in for the rwkv.cpp hot path (librwkv.so + its Python host mirror).

The directory name contains a dot, so import it through `__graft_entry__.load_package()` (importlib by path) or add the
repo root to sys.path and use `importlib.import_module("rwkv.cpp_amd")` is NOT possible; see __graft_entry__.py.
"""
from .rwkv_cpp import (  # noqa: F401
    HOOKS_LIB_PATH,
    LIB_PATH,
    RWKVContext,
    RWKVModel,
    RWKVSharedLibrary,
    build_library,
    load_rwkv_shared_library,
)
