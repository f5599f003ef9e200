"""Build libdws.so (all HIP kernels + the C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``diffwave-sashimi_amd/libdws.so`` is git-ignored but travels to the
GPU box with the tree.
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdws.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "dws.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source into one shared library.  Raises on failure."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-o", LIB] + sources() + ["-L/opt/rocm/lib", "-lhipfft", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout[-4000:] + proc.stderr[-8000:])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
