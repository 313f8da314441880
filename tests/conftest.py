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
    """Built artefacts are git-ignored; build them on first use so a fresh checkout can run the suite.  The oracle's C
    restatement needs gcc only; the HIP library needs hipcc -- without it the tests that load the library fail loudly
    (there is no CPU fallback), the rest of the CPU suite still runs."""
    import shutil
    import subprocess
    hip_so = os.path.join(REPO, "probabilistic-depth_amd", "libpdepth_hip.so")
    ora_so = os.path.join(REPO, "oracle", "libpdepth_oracle.so")
    if not os.path.exists(ora_so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    if not os.path.exists(hip_so):
        if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) or shutil.which("hipcc"):
            subprocess.check_call(["make", "-C", os.path.join(REPO, "probabilistic-depth_amd", "csrc"), "-j4"])
        else:
            warnings.warn("hipcc not found: libpdepth_hip.so was not built, tests that load it will fail")


_ensure_native_built()
