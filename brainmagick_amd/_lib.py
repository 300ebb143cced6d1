"""ctypes binding of ``libbmhip.so`` (the C-ABI declared in ``include/bm_hip.h``).

The prototypes are parsed from the header itself so that the binding can never drift from the
declared ABI.  There is NO fallback: if the shared library (or a symbol) is missing, loading raises
and every op of the hot path raises with it.
"""
import ctypes
import os
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HEADER = ROOT.parent / "include" / "bm_hip.h"
LIB_PATH = Path(os.environ["BM_HIP_LIB"]) if os.environ.get("BM_HIP_LIB") else ROOT / "libbmhip.so"
CSRC = ROOT / "csrc"

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
           "double": ctypes.c_double}


class BmHipError(RuntimeError):
    pass


def parse_header(path: Path = HEADER):
    """Returns {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    text = path.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*(const char\*|int|long)\s+(bm_\w+)\s*\(([^;]*?)\)\s*;", text,
                         flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        restype = ctypes.c_char_p if ret == "const char*" else _CTYPES[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for arg in args.split(","):
                arg = arg.strip()
                if "*" in arg:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(arg.split("*")[-1].strip())
                else:
                    typ, nm = arg.rsplit(" ", 1)
                    argtypes.append(_CTYPES[typ.replace("const ", "").strip()])
                    argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into the in-tree libbmhip.so (hipcc cross-compiles on CPU)."""
    sources = sorted(CSRC.glob("*.hip"))
    deps = sources + sorted(CSRC.glob("*.h"))
    if not force and LIB_PATH.exists() and \
            LIB_PATH.stat().st_mtime >= max(p.stat().st_mtime for p in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
    # one object per translation unit, compiled concurrently (the wide-tile kernels take ~40 s each), then linked.
    # Objects are kept under build/obj (git-ignored): a unit is recompiled when it, or any header, is newer than
    # its object -- `force` recompiles everything (what __graft_entry__.build() does).
    import concurrent.futures
    objdir = ROOT.parent / "build" / "obj"
    objdir.mkdir(parents=True, exist_ok=True)
    newest_header = max([p.stat().st_mtime for p in CSRC.glob("*.h")] + [0.0])

    def compile_one(src: Path):
        obj = objdir / (src.stem + ".o")
        if not force and obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, newest_header):
            return obj
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise BmHipError(f"hipcc failed on {src.name}:\n{proc.stdout}\n{proc.stderr}")
        return obj
    workers = max(1, min(len(sources), os.cpu_count() or 1, 8))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        objects = list(pool.map(compile_one, sources))
    for stale in objdir.glob("*.o"):            # a source file that went away
        if stale.stem not in {src.stem for src in sources}:
            stale.unlink()
    cmd = [hipcc, *flags, "-shared", *map(str, objects), "-o", str(LIB_PATH)]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise BmHipError(f"hipcc link failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


_lib = None
_protos = None


def lib():
    """The loaded library with typed prototypes; raises BmHipError if it cannot be loaded."""
    global _lib, _protos
    if _lib is None:
        if not LIB_PATH.exists():
            raise BmHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
        handle = ctypes.CDLL(str(LIB_PATH))
        protos = parse_header()
        for name, (restype, argtypes, _) in protos.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as exc:
                raise BmHipError(f"libbmhip.so does not export {name} declared in bm_hip.h") from exc
            fn.restype = restype
            fn.argtypes = argtypes
        _lib, _protos = handle, protos
    return _lib


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib().bm_last_error()
        raise BmHipError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")
