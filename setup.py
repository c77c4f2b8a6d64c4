"""Build + install: compiles the gfx950 HIP library and the host-only torch extension modules with
di-hpc_amd/build.py (hipcc / g++), then lays them out next to the ``hpc_rll`` package so that the modules' rpath
``$ORIGIN/hpc_rll/_lib`` resolves in site-packages exactly as in the source tree.  See pyproject.toml."""
import os
import shutil
import sys

from setuptools import Extension, setup
from setuptools.command.build_ext import build_ext

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "di-hpc_amd")
MODULES = ["hpc_rl_utils", "hpc_torch_utils_network", "hpc_models"]


class BuildNative(build_ext):
    def run(self):
        sys.path.insert(0, PKG)
        import build as native_build   # di-hpc_amd/build.py
        native_build.build(force=False, verbose=True)
        for ext in self.extensions:     # prebuilt by build.py: copy under the interpreter's extension file name
            dst = self.get_ext_fullpath(ext.name)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copy2(os.path.join(PKG, ext.name + ".so"), dst)
        lib_root = os.path.dirname(self.get_ext_fullpath(MODULES[0]))
        for sub, src in (("_lib", os.path.join(PKG, "hpc_rll", "_lib", "libhpc_rll_hip.so")),
                         ("include", os.path.join(ROOT, "include", "hpc_rll_hip.h"))):
            d = os.path.join(lib_root, "hpc_rll", sub)
            os.makedirs(d, exist_ok=True)
            shutil.copy2(src, d)


setup(
    name="di-hpc-amd",
    version="0.2.0",
    description="MI355X (gfx950) native operator library behind the hpc_rll.rl_utils / hpc_rll.torch_utils API",
    python_requires=">=3.9",
    install_requires=["torch"],
    package_dir={"": "di-hpc_amd"},
    packages=["hpc_rll", "hpc_rll.rl_utils", "hpc_rll.torch_utils", "hpc_rll.torch_utils.network", "di_hpc_amd"],
    ext_modules=[Extension(m, sources=[]) for m in MODULES],
    cmdclass={"build_ext": BuildNative},
    zip_safe=False,
)
