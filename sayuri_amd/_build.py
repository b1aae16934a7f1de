"""Build the native libraries in-tree (they travel to the GPU box with the snapshot).

    sayuri_amd/lib/libsayuri_hip.so   hipcc --offload-arch=gfx950 (kernels + C-ABI)
    sayuri_amd/lib/libsayuri_host.so  g++ (weights loader, HipForwardPipe, ctypes wrapper)
    oracle/libsayuri_oracle.so        gcc (CPU checker, test infrastructure)
    oracle/_ref/libsayuri_ref.so      g++ on the reference's own sources, only where
                                      /root/reference exists (dev container)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sayuri_amd")
LIB = os.path.join(PKG, "lib")
HIP_SRC = os.path.join(PKG, "csrc", "hip")
HOST_SRC = os.path.join(PKG, "csrc", "host")
ENGINE_SRC = os.path.join(PKG, "csrc", "engine")
# The evaluation post-processing and the tree-search arithmetic are built with the floating-point flags of the
# reference's own build (CMakeLists.txt:192-203: -O3 -ffast-math, here with x86-64-v3 instead of -march=native) so
# that a fixed-seed search rounds like the reference binary does.
FASTMATH_UNITS = {"network.cc", "tree.cc", "search.cc", "search_params.cc"}
HIP_SO = os.path.join(LIB, "libsayuri_hip.so")
HOST_SO = os.path.join(LIB, "libsayuri_host.so")


def _digest(sources, extra=()) -> str:
    """sha256 over the CONTENTS of the sources and whatever else decides the output (command lines, options)."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(sources):
        h.update(os.path.basename(s).encode() + b"\0")
        with open(s, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _stale(target: str, sources, extra=()) -> str:
    """"" when `target` exists and its stamp names this digest, else the digest to stamp after the rebuild.  Contents, not
    mtimes: a checkout, a copy to another box or a touched file neither forces nor hides a rebuild; a changed option
    (SAYURI_TOWER_PAD, SAYURI_EXPERIMENTS ...) does force one."""
    d = _digest(sources, extra)
    try:
        with open(target + ".stamp") as f:
            if os.path.exists(target) and f.read().strip() == d:
                return ""
    except OSError:
        pass
    return d


def _stamp(target: str, digest: str) -> None:
    with open(target + ".stamp", "w") as f:
        f.write(digest + "\n")


def _files(d: str, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


_TOOLCHAIN = None


def _toolchain() -> str:
    """What `hipcc --version` says (and where ROCm is): part of every digest of a device-side output.  tower_seam.py
    depends on the exact register assignment hipcc produced; a stamped library must not survive a ROCm upgrade."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            v = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=120).stdout
        except Exception as e:  # noqa: BLE001
            v = "hipcc --version failed: %r" % (e,)
        _TOOLCHAIN = v.strip() + "|" + os.environ.get("ROCM_PATH", "")
    return _TOOLCHAIN


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _llvm_bin() -> str:
    """The LLVM tools that belong to the hipcc in use (clang as assembler, ld.lld): next to it, or under ROCM_PATH."""
    hip = os.path.realpath(_hipcc())
    roots = [os.path.dirname(os.path.dirname(hip)), os.environ.get("ROCM_PATH", ""), "/opt/rocm"]
    for r in roots:
        for sub in ("lib/llvm/bin", "llvm/bin"):
            d = os.path.join(r, sub) if r else ""
            if d and os.path.exists(os.path.join(d, "clang")) and os.path.exists(os.path.join(d, "ld.lld")):
                return d
    raise RuntimeError("clang / ld.lld of the ROCm LLVM not found (looked next to %s and under ROCM_PATH)" % hip)


def _seam_options():
    # where the plain body starts: 256-byte boundary + 32 bytes, the best of seven placements measured on two boxes
    # (NOTES.md, Kernel 1c); SAYURI_TOWER_ALIGN / SAYURI_TOWER_PAD build the others
    return ((["--inv"] if os.environ.get("SAYURI_TOWER_INV") else []) +
            (["--sleep=" + os.environ["SAYURI_TOWER_SLEEP"]] if os.environ.get("SAYURI_TOWER_SLEEP") else []) +
            ["--align=" + os.environ.get("SAYURI_TOWER_ALIGN", "8"), "--pad=" + os.environ.get("SAYURI_TOWER_PAD", "32")])


def _write_blob(objdir: str, hsaco: str | None, verbose: bool = False) -> str:
    """tower_blob.o: the code object as the byte array `sayuri_tower_hsaco` + its length `sayuri_tower_hsaco_size`.
    hsaco = None writes the EMPTY blob (size 0): libsayuri_hip.so then carries no persistent tower kernel, the engine says so
    once at creation and runs one launch per layer (engine.hip: load_tower_module, tower_ok)."""
    stub, blob = os.path.join(objdir, "tower_blob.S"), os.path.join(objdir, "tower_blob.o")
    with open(stub, "w") as f:
        f.write('\t.section .rodata\n\t.globl sayuri_tower_hsaco\n\t.type sayuri_tower_hsaco,@object\n\t.balign 4096\n'
                'sayuri_tower_hsaco:\n' +
                ('\t.incbin "%s"\n' % os.path.basename(hsaco) if hsaco else '\t.quad 0\n') +
                '.Lsayuri_tower_hsaco_end:\n\t.size sayuri_tower_hsaco, .-sayuri_tower_hsaco\n'
                '\t.globl sayuri_tower_hsaco_size\n\t.type sayuri_tower_hsaco_size,@object\n\t.balign 8\n'
                'sayuri_tower_hsaco_size:\n\t.quad ' + ('.Lsayuri_tower_hsaco_end - sayuri_tower_hsaco' if hsaco else '0') + '\n'
                '\t.size sayuri_tower_hsaco_size, 8\n'
                '\t.section .note.GNU-stack,"",@progbits\n')
    cmd = ["gcc", "-c", os.path.basename(stub), "-o", os.path.basename(blob)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=objdir)  # .incbin is looked up relative to the working directory: no absolute path in the stub
    return blob


def tower_blob_from_asm(asm: str, objdir: str, verbose: bool = False):
    """hipcc's assembly of tower.hip -> (tower_blob.o, seamed).  tower_seam.py closes the layer loop in it and checks, line by
    line, that the assembly looks like what it was written against (anchors, clobber ranges, the FC body's registers: it
    depends on hipcc's register assignment, validated on ROCm 7.2's hipcc).  When it REJECTS the assembly -- another compiler
    version moved an anchor -- the build goes on without the persistent kernel: the empty blob is linked, the engine reports
    the fallback at creation and runs the same convolutions one launch per layer (tower_ok() is false).  SAYURI_TOWER_REQUIRED=1
    turns the rejection back into a build failure (development)."""
    seam = os.path.join(HIP_SRC, "tower_seam.py")
    seamed = os.path.join(objdir, "tower_seamed.s")
    elf, hsaco = os.path.join(objdir, "tower_dev.o"), os.path.join(objdir, "tower.hsaco")
    cmd = [sys.executable, seam, asm, seamed] + _seam_options()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        msg = (r.stderr or r.stdout).strip().splitlines()[-1:] or ["(no message)"]
        if os.environ.get("SAYURI_TOWER_REQUIRED"):
            raise RuntimeError("tower_seam.py rejected the compiler's assembly: " + msg[0])
        print("[sayuri build] WARNING: tower_seam.py rejected hipcc's assembly of tower.hip (%s).\n"
              "[sayuri build]          libsayuri_hip.so is built WITHOUT the persistent tower kernel; the engine will run one launch "
              "per layer.\n[sayuri build]          The seam is validated on ROCm 7.2's hipcc; SAYURI_TOWER_REQUIRED=1 makes this an error."
              % msg[0], file=sys.stderr)
        return _write_blob(objdir, None, verbose), False
    for cmd in ([os.path.join(_llvm_bin(), "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", seamed, "-o", elf],
                [os.path.join(_llvm_bin(), "ld.lld"), "-shared", elf, "-o", hsaco]):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return _write_blob(objdir, hsaco, verbose), True


def build_tower_blob(force: bool = False, verbose: bool = False) -> str:
    """The persistent tower kernels (csrc/hip/conv_tower.h): tower.hip -> gfx950 assembly -> tower_seam.py closes the layer
    loop in it -> code object -> a host object that carries it as the byte array `sayuri_tower_hsaco`."""
    objdir = os.path.join(LIB, "obj")
    os.makedirs(objdir, exist_ok=True)
    blob = os.path.join(objdir, "tower_blob.o")
    seam = os.path.join(HIP_SRC, "tower_seam.py")
    srcs = [os.path.join(HIP_SRC, f) for f in ("tower.hip", "conv_tower.h", "conv_board.h", "conv_glds.h", "conv_mfma.h",
                                               "small_ops.h", "common.h")] + [seam]
    hip_flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-w"]
    if os.environ.get("SAYURI_EXPERIMENTS"):
        hip_flags.insert(0, "-DSAYURI_EXPERIMENTS")
    extra = hip_flags + _seam_options() + [_toolchain(), "required" if os.environ.get("SAYURI_TOWER_REQUIRED") else ""]
    digest = _stale(blob, srcs, extra)
    if not (force or digest):
        return blob
    digest = digest or _digest(srcs, extra)
    asm = os.path.join(objdir, "tower.s")
    cmd = [_hipcc()] + hip_flags + [os.path.join(HIP_SRC, "tower.hip"), "-o", asm]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    tower_blob_from_asm(asm, objdir, verbose)
    _stamp(blob, digest)
    return blob


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    srcs = _files(HIP_SRC, (".hip", ".h", ".py")) + [os.path.join(ROOT, "include", "sayuri_hip.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w"]
    if os.environ.get("SAYURI_EXPERIMENTS"):  # in-kernel timelines and forced variants (measuring builds only)
        flags.insert(0, "-DSAYURI_EXPERIMENTS")
    extra = flags + _seam_options() + [_toolchain()]
    digest = _stale(HIP_SO, srcs, extra)
    if force or digest:
        digest = digest or _digest(srcs, extra)
        blob = build_tower_blob(force, verbose)
        obj = os.path.join(LIB, "obj", "engine_hip.o")
        cmd = [_hipcc()] + flags + ["-c", os.path.join(HIP_SRC, "engine.hip"), "-o", obj]
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", obj, blob, "-o", HIP_SO]
        for c in (cmd, link):
            if verbose:
                print(" ".join(c), file=sys.stderr)
            subprocess.check_call(c)
        _stamp(HIP_SO, digest)
    return HIP_SO


def build_host(force: bool = False, verbose: bool = False) -> str:
    """g++ every host/ and engine/ translation unit to an object (rebuilt only when it or a header
    changed), then link libsayuri_host.so against libsayuri_hip.so."""
    from concurrent.futures import ThreadPoolExecutor
    build_hip(force=False, verbose=verbose)
    objdir = os.path.join(LIB, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = _files(HOST_SRC, (".h",)) + _files(ENGINE_SRC, (".h",)) + [os.path.join(ROOT, "include", f)
                                                                          for f in os.listdir(os.path.join(ROOT, "include"))]
    jobs, objs, stamps = [], [], []
    hdr_digest = _digest(headers)
    for src in _files(HOST_SRC, (".cc",)) + _files(ENGINE_SRC, (".cc",)):
        obj = os.path.join(objdir, os.path.basename(os.path.dirname(src)) + "_" + os.path.basename(src)[:-3] + ".o")
        objs.append(obj)
        opt = ["-O2"]
        if os.path.basename(src) in FASTMATH_UNITS:
            opt = ["-O3", "-ffast-math", "-march=x86-64-v3"]
        flags = ["-std=c++17"] + opt + ["-fPIC", "-Wall", "-Wextra"]
        digest = _stale(obj, [src], flags + [hdr_digest])
        if force or digest:
            stamps.append((obj, digest or _digest([src], flags + [hdr_digest])))
            jobs.append(["g++"] + flags + ["-I" + HOST_SRC, "-I" + ENGINE_SRC, "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
    if jobs:
        if verbose:
            for j in jobs:
                print(" ".join(j), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(16, len(jobs))) as ex:
            list(ex.map(subprocess.check_call, jobs))
        for obj, digest in stamps:
            _stamp(obj, digest)
    cmd = ["g++", "-shared", "-o", HOST_SO] + objs + ["-L" + LIB, "-lsayuri_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lz"]
    link_extra = [" ".join(os.path.relpath(c, ROOT) if os.path.isabs(c) else c for c in cmd)]  # the link line decides the output too
    link_digest = _stale(HOST_SO, objs, link_extra)
    if jobs or force or link_digest:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _stamp(HOST_SO, _digest(objs, link_extra))
    return HOST_SO


def build_oracle(force: bool = False, verbose: bool = False) -> None:
    odir = os.path.join(ROOT, "oracle")
    out = None if verbose else subprocess.DEVNULL
    always = ["-B"] if force else []
    subprocess.check_call(["make", "-C", odir] + always + ["port"], stdout=out)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", odir, "-j8"] + always + ["ref"], stdout=out, stderr=out)


def build_all(force: bool = False, verbose: bool = False) -> None:
    """force=True (or SAYURI_BUILD_FORCE=1) recompiles everything from source whatever is in the tree."""
    force = force or bool(os.environ.get("SAYURI_BUILD_FORCE"))
    build_hip(force, verbose)
    build_host(force, verbose)
    build_oracle(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
