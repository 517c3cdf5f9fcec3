"""pytest configuration: markers, repo root on sys.path, and a one-time native build when the
in-tree libraries are missing (fresh checkout: *.so files are git-ignored)."""
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    lib = ROOT / "quantum_simulations_amd" / "libqsim_hip.so"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not lib.exists() and Path(hipcc).exists():
        subprocess.run(["make", "-C", str(ROOT / "quantum_simulations_amd" / "csrc"), f"HIPCC={hipcc}"],
                       check=False, capture_output=True)
    if not (ROOT / "oracle" / "liborc.so").exists() and shutil.which("gcc"):
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=False, capture_output=True)
