import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the product library + CLI once when hipcc is
    around (cross-compiles gfx950 without a GPU).  On the GPU box the prebuilt files travel with the snapshot."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "vicalib_amd", "libvicalib_amd.so")
    cli = os.path.join(ROOT, "vicalib_amd", "vicalib")
    gen = os.path.join(ROOT, "vicalib_amd", "libvicalib_synth.so")        # the C++ problem generator (test / bench infrastructure)
    if os.path.exists(lib) and os.path.exists(cli) and os.path.exists(gen):
        return
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        return          # the tests that need the library will say so
    subprocess.check_call(["bash", os.path.join(ROOT, "vicalib_amd", "csrc", "build.sh")])
