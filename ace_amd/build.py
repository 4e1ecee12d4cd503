"""In-tree build of libace_sfno.so for gfx950 (hipcc cross-compiles without a GPU).

One object per source file, compiled in parallel and cached by modification time (csrc/build/), then one link:
an edit of a single kernel file rebuilds in well under a minute."""

import concurrent.futures
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libace_sfno.so")
SOURCES = ["kernels.hip", "fft.hip", "strip.hip", "strip_fold.hip", "conv_ws.hip", "conv_wl.hip", "dhconv_strip.hip", "cln_mfma.hip", "physics.hip", "healpix.hip", "capi.hip", "tables.cpp"]
HEADERS = ["kernels.h", "strip_common.h", "strip_pack.h", "ws_plan.h", "pack_frag.h", "dhconv_units.h", "tuning_guard.h", "small_fft.h", "tables.h", os.path.join("..", "..", "include", "ace_sfno.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libace_sfno.so)")


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _header_time() -> float:
    return max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)


def source_sha256() -> str:
    """sha256 over the kernel sources, headers and compile flags: identifies the CODE a library was built from (hipcc's output is
    not bit-reproducible across object paths, so profiles/ stamps its counter files with this as well as with the .so's hash)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _compile(hipcc: str, src: str, obj: str, extra, verbose: bool) -> None:
    cmd = [hipcc] + FLAGS + list(extra) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)


def build(force: bool = False, verbose: bool = False, out: str = LIB, extra=(), objdir: str = OBJ) -> str:
    """Compile the HIP library for gfx950; returns the path of the .so.  `extra`: additional compiler flags (-D...) for a
    variant build, which then needs its own `out` and `objdir`."""
    if out == LIB and extra:
        raise ValueError("the shipped library is built with the default flags only; a variant (extra flags) needs its own `out` and `objdir`")
    if extra and not any(f == "-DACE_MEASUREMENT_SWITCHES" for f in extra):
        extra = list(extra) + ["-DACE_MEASUREMENT_SWITCHES"]   # csrc/tuning_guard.h: any -D of a tuning / ablation / trace macro has to say so
    if not force and out == LIB and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    for stale in os.listdir(objdir):   # objects of sources that no longer exist (removed kernels) do not linger
        if stale.endswith(".o") and stale[:-2] not in SOURCES:
            os.remove(os.path.join(objdir, stale))
    ht = _header_time()
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(src), ht):
            jobs.append((src, obj))
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for f in [ex.submit(_compile, hipcc, src, obj, extra, verbose) for src, obj in jobs]:
            f.result()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    import sys

    print(build(force="--force" in sys.argv, verbose=True))
