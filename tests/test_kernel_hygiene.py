"""The hand-scheduled K loops keep LDS fragments in registers that an inline-asm ds_read is still filling
(sayuri_amd/csrc/hip/conv_board.h, conv_glds.h): a compiler spill or copy of such a register between the read and its
s_waitcnt stores the OLD contents -- wrong results, not just slow ones (the persistent tower).  This test
disassembles the gfx950 code object inside the built library and asserts that no scratch access sits inside the MFMA
stream of the board kernels, and that the dominant kernel has no vector spill at all.  CPU-only: it reads the .so."""
import os
import re
import struct
import subprocess

import pytest

from sayuri_amd import _build

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_object(so_path: str, tmp_path) -> str:
    blob = open(so_path, "rb").read()
    at = blob.find(MAGIC)
    assert at >= 0, "no offload bundle in the library"
    n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
    pos = at + len(MAGIC) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24:pos + 24 + tlen].decode()
        pos += 24 + tlen
        if "gfx950" in triple:
            out = os.path.join(str(tmp_path), "dev.co")
            open(out, "wb").write(blob[at + off:at + off + size])
            return out
    raise AssertionError("no gfx950 code object in the bundle")


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(READELF)), reason="needs the ROCm llvm tools")
def test_no_spill_inside_the_mfma_streams(tmp_path):
    so = _build.HIP_SO
    if not os.path.exists(so):
        _build.build_hip()
    co = device_code_object(so, tmp_path)
    asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    kernels = {}
    name = None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip():
            kernels[name].append(line.strip())
    checked = 0
    for name, body in kernels.items():
        if not re.search(r"conv_board(_se)?_kernel|conv_glds_kernel|head_board_kernel", name):
            continue
        mf = [i for i, ins in enumerate(body) if ins.startswith("v_mfma")]
        if not mf:
            continue
        inside = [ins for ins in body[mf[0]:mf[-1] + 1] if ins.startswith("scratch_") or ins.startswith("buffer_store_dword v") and "offen" in ins]
        assert not inside, f"{name}: {len(inside)} scratch accesses inside the MFMA stream, e.g. {inside[:3]}"
        # the epilogue's register-form residual pieces (gload16_s: global_load_dwordx4 with a scalar base) are in flight until
        # the next vmcnt(0): no spill traffic in that window either
        for i, ins in enumerate(body):
            if "ELb1EE" in name:  # the timeline build (SAYURI_BOARD_DBG) is a measuring tool, not a product path
                break
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[", ins)
            if not m:
                continue
            lo, hi = int(m.group(1)), int(m.group(2))
            for later in body[i + 1:]:
                if later.startswith("s_waitcnt") and "vmcnt(0)" in later:
                    break
                if later.startswith("s_endpgm"):
                    break
                # the registers the load is still writing may be neither saved nor handed out again
                sm = re.match(r"scratch_(load|store)_dword\w* (?:v(\d+)|v\[(\d+):(\d+)\])", later)
                if sm:
                    a = int(sm.group(2) or sm.group(3))
                    b = int(sm.group(2) or sm.group(4))
                    assert b < lo or a > hi, f"{name}: '{later[:60]}' touches v[{lo}:{hi}] while a residual piece is in flight to it"
        checked += 1
    assert checked >= 6, f"only {checked} board / glds kernels found in the code object"
    # the dominant kernel: no vector spill anywhere
    notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True, check=True).stdout
    blocks = notes.split(".agpr_count")
    hit = [b for b in blocks if "conv_board_kernelILi4ELb0" in b]
    assert hit, "conv_board_kernel<4,false> not found in the metadata"
    m = re.search(r"\.vgpr_spill_count:\s+(\d+)", hit[0])
    assert m and int(m.group(1)) == 0, f"conv_board_kernel<4,false> spills {m.group(1) if m else '?'} vector registers"
