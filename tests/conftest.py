import os
import sys
import warnings

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

warnings.filterwarnings("ignore", message="Default grid_sample")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_native_built():
    """Built artefacts are git-ignored; build them on first use so a fresh checkout can run the suite."""
    import subprocess
    hip_so = os.path.join(REPO, "probabilistic-depth_amd", "libpdepth_hip.so")
    ora_so = os.path.join(REPO, "oracle", "libpdepth_oracle.so")
    if not os.path.exists(hip_so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "probabilistic-depth_amd", "csrc"), "-j4"])
    if not os.path.exists(ora_so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])


_ensure_native_built()
