"""Build libdws.so (all HIP kernels + the C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``diffwave-sashimi_amd/libdws.so`` is git-ignored but travels to the
GPU box with the tree.  Each ``csrc/*.hip`` is compiled to its own object under
``csrc/build/`` (in parallel, only when it or a header changed), then linked.
"""
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libdws.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# Per-file additions.  fftconv_kernels: the SLP vectoriser turns the complex butterflies into v_pk_*_f32 plus ~450
# v_mov shuffles per kernel; a packed fp32 instruction costs 4.3 cycles against 2.35-2.6 for a scalar one
# (tools/valu_rate.hip), so with the moves the scalar form is shorter -- and spill-free at 128 VGPRs.
FILE_FLAGS = {"fftconv_kernels": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "dws.h")]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _extra_flags(stem):
    extra = os.environ.get("DWS_HIPCC_FLAGS_" + stem)      # experiments: override a file's extra flags
    return FILE_FLAGS.get(stem, []) if extra is None else extra.split()


def _compile_cmd(hipcc, src):
    stem = os.path.basename(src)[:-4]
    return [hipcc] + FLAGS + _extra_flags(stem) + ["-c", src, "-o", _obj(src)]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _cmd_key(src, hdr_digest=None):
    """What an object was built FROM: its flags (no paths: the tree is copied to another directory on the GPU box) and
    the CONTENTS of the source and of every header -- not modification times, which a copied tree does not keep in any
    useful order (a stale libdws.so travelling with newer sources would otherwise be reused)."""
    hdr_digest = hdr_digest or _digest(_headers())
    return " ".join(FLAGS + _extra_flags(os.path.basename(src)[:-4])) + "\n" + _digest([src]) + "\n" + hdr_digest


def _cmd_changed(hipcc, src, hdr_digest=None):
    """An object is stale when its flags (FLAGS, FILE_FLAGS or a DWS_HIPCC_FLAGS_<stem> experiment), its source or a
    header differ from what the sidecar next to it recorded, or when there is no object."""
    try:
        return not os.path.exists(_obj(src)) or open(_obj(src) + ".cmd").read() != _cmd_key(src, hdr_digest)
    except OSError:
        return True


def _lib_key(keys):
    return hashlib.sha256("\n".join(keys).encode()).hexdigest()


def needs_build():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hd = _digest(_headers())
    if not os.path.exists(LIB) or any(_cmd_changed(hipcc, s, hd) for s in sources()):
        return True
    try:   # the library was linked from exactly these objects
        return open(LIB + ".key").read() != _lib_key([_cmd_key(s, hd) for s in sources()])
    except OSError:
        return True


def build(force=False, verbose=False):
    """Compile every .hip source into one shared library.  Raises on failure."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hd = _digest(_headers())
    todo = [s for s in sources() if force or _cmd_changed(hipcc, s, hd)]

    def compile_one(src):
        cmd = _compile_cmd(hipcc, src)
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(_obj(src) + ".cmd"):
            os.remove(_obj(src) + ".cmd")
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, proc.stdout[-4000:], proc.stderr[-8000:]))
        with open(_obj(src) + ".cmd", "w") as f:
            f.write(_cmd_key(src, hd))

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in sources()] + \
          ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    if os.path.exists(LIB + ".key"):
        os.remove(LIB + ".key")
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n" + proc.stdout[-4000:] + proc.stderr[-8000:])
    with open(LIB + ".key", "w") as f:
        f.write(_lib_key([_cmd_key(s, hd) for s in sources()]))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose="-v" in os.sys.argv))
