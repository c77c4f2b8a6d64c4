"""``di_hpc_amd`` -- the importable name of this distribution (MI355X-native ``hpc_rll`` operator library).

The drop-in API lives under the reference's own module names, which is the point of a drop-in:
``hpc_rll.rl_utils.*``, ``hpc_rll.torch_utils.network.*`` (Python) and the compiled extension modules ``hpc_rl_utils``,
``hpc_torch_utils_network``, ``hpc_models``.  This module only answers "where is the native code":

    di_hpc_amd.get_library()   path of libhpc_rll_hip.so (the torch-free C ABI)
    di_hpc_amd.get_include()   directory holding hpc_rll_hip.h (the C ABI header)
"""
import os

__version__ = "0.2.0"

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hpc_rll")


def get_library() -> str:
    return os.path.join(_PKG, "_lib", "libhpc_rll_hip.so")


def get_include() -> str:
    """Installed tree: hpc_rll/include (shipped inside the package); source tree: <repo>/include."""
    inst = os.path.join(_PKG, "include")
    if os.path.exists(os.path.join(inst, "hpc_rll_hip.h")):
        return inst
    return os.path.join(os.path.dirname(os.path.dirname(_PKG)), "include")
