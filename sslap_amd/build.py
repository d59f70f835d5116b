"""Build libmisslap.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmisslap.so")
LIB_DIAG = os.path.join(PKG, "libmisslap_diag.so")  # diagnostics build (-DMISSLAP_DIAG), tools/ only
SOURCES = ["misslap.hip", "device_common.hpp", "kernels_round.hpp", "kernels_tail.hpp", "kernels_check.hpp", "kernels_debug.hpp", "kernels_tiled.hpp", "host_comm.hpp", "kernels_matching.hpp",
           "kernels_ingest.hpp", "host_matching.hpp", "abi_v1.hpp", "host_base.hpp", "host_batch.hpp", "host_cache.hpp", "host_rounds.hpp", "host_create.hpp",
           "abi_matching.hpp", "abi_util.hpp", "abi_comm.hpp", "abi_batch.hpp", "abi_diag.hpp", os.path.join("..", "..", "include", "misslap.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-honor-nans", "-std=c++17", "-shared", "-fPIC",
         "-fvisibility=hidden", "-Wall", "-Wextra"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmisslap.so can only be built with the ROCm toolchain")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=False):
    """Compile sslap_amd/csrc/misslap.hip -> sslap_amd/libmisslap.so.  Returns the library path."""
    if force or stale():
        cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, "misslap.hip"), "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def build_diag(verbose=False):
    """The same sources with -DMISSLAP_DIAG (ablation kernels + misslap_debug_time_bid, include/misslap_diag.h)."""
    cmd = [hipcc()] + FLAGS + ["-DMISSLAP_DIAG", os.path.join(CSRC, "misslap.hip"), "-o", LIB_DIAG]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_DIAG


if __name__ == "__main__":
    import sys
    print(build_diag(verbose=True) if "diag" in sys.argv[1:] else build(force=True, verbose=True))
