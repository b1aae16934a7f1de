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


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs the ROCm llvm tools")
def test_persistent_tower_code_object():
    """The persistent tower launch (conv_tower.h) is hipcc's single-layer convolution kernel with the layer loop closed in
    assembly and the SE unit put in as generated assembly around a compiled FC body (tower_seam.py).  The code object that
    ships inside libsayuri_hip.so must (i) be the one the build produced, (ii) carry the entry stub, the dispatch, the SE hook
    (pooling -> FC body -> gate, all behind the K loop and in front of the epilogue) and one seam, (iii) use NO scratch at all --
    the SE unit of rounds 2-3 parked six accumulator tiles there -- and no AGPR inside the FC body, (iv) start the convolution
    body where the build places it, and (v) declare 160 KiB of static LDS and no private segment in the launch descriptor."""
    so = _build.HIP_SO
    if not os.path.exists(so):
        _build.build_hip()
    hsaco = os.path.join(_build.LIB, "obj", "tower.hsaco")
    assert os.path.exists(hsaco), "build_tower_blob() leaves the code object next to the objects"
    blob = open(hsaco, "rb").read()
    assert blob in open(so, "rb").read(), "libsayuri_hip.so does not embed lib/obj/tower.hsaco"
    asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", hsaco], capture_output=True, text=True, check=True).stdout
    order, sections, address = [], {}, {}
    name = None
    for line in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            name = m.group(2)
            order.append(name)
            sections[name] = []
            address[name] = int(m.group(1), 16)
        elif name and line.strip():
            sections[name].append(line.split("//")[0].strip())
    assert not [i for body in sections.values() for i in body if i.startswith("scratch_")], "scratch access in the tower code object"
    for w in (4, 2):
        entry = f"_ZN6sayuri17conv_tower_kernelILi{w}EEEvPKNS_10TowerLayerE"
        assert entry in sections and f"tower{w}_dispatch" in sections, "entry stub / dispatch missing"
        stub = sections[entry]
        assert stub[0].startswith("s_load_dwordx2") and "s[0:1]" in stub[0], stub[0]
        disp = sections[f"tower{w}_dispatch"]
        assert any(i.startswith("s_mov_b64 exec, -1") for i in disp) and any(i.startswith("v_mbcnt_hi") for i in disp)
        assert disp[0].startswith("s_load_dword ") and disp[0].endswith("0xc"), "the dispatch parks has_se: " + disp[0]
        # placement (tower_seam.py --align=8 --pad=N, N = 32 unless the build was told otherwise): measured, DESIGN.md Kernel 1c
        assert address[f"tower{w}_body_conv"] % 256 == int(os.environ.get("SAYURI_TOWER_PAD", "32")), hex(address[f"tower{w}_body_conv"])
        # the convolution body up to the FC's return label: K loop, then the hook's pooling (accumulators read in place:
        # v_accvgpr_read, DPP row reductions, the partials written to LDS), then the far jump into the FC body
        head = sections[f"tower{w}_body_conv"]
        mf = [i for i, x in enumerate(head) if x.startswith("v_mfma")]
        assert len(mf) > 100, "no MFMA stream in the convolution body"
        nop = next(i for i in range(mf[-1], len(head)) if head[i].startswith("s_nop 15"))
        after = head[nop:]  # the K loop's exit: two s_nop 15 (the last MFMAs retire), then the hook
        assert after[1].startswith("s_nop 15") and after[2].startswith("s_cmp_eq_u32") and after[3].startswith("s_cbranch_scc1"), after[:4]
        assert any("row_ror:8" in x for x in after) and any(x.startswith("ds_write_b128") for x in after), "pooling missing behind the K loop"
        assert after[-1].startswith("s_setpc_b64"), "far jump into the FC body: " + after[-1]
        assert not [x for x in after if x.startswith("v_mfma")]
        # behind the return label: the gate in place (fused multiply-adds between accvgpr reads and writes), then the compiled
        # epilogue and the seam: every exit goes through  vmcnt(0) -> s_barrier -> (end | next element)
        tail = sections[f"tower{w}_se_return"]
        first_store = next(i for i, x in enumerate(tail) if x.startswith("global_store"))
        gate = tail[:first_store]
        assert sum(x.startswith("v_pk_fma_f32") for x in gate) >= 2 * w * 12, "the gate's fused multiply-adds"
        assert sum(x.startswith("v_accvgpr_write") for x in gate) >= 4 * min(32, w * 12), "gated tiles go back to their AGPRs"
        # ... then the generated epilogue for Mish layers with computed table entries (one v_fma_mix_f32 per residual value, the
        # Mish of mish2 operation for operation, 16-byte stores) which leaves by a far jump to the seam, then the compiled epilogue
        gen = tail[first_store - 200:]
        assert any(x.startswith("v_fma_mix_f32") for x in tail) and any(x.startswith("v_cvt_pk_f16_f32") for x in gen)
        # its three activations: Mish (per store 8 x v_exp_f32 + 8 x v_rcp_f32), ReLU (compare + select), none
        stores = sum(x.startswith("global_store_dwordx4") for x in gen[:gen.index(next(x for x in gen if x.startswith("s_setpc_b64")))])
        assert stores == 6 * 12 * (w // 2), stores
        assert sum(x.startswith("v_exp_f32") for x in gen) >= 8 * stores // 3 and sum(x.startswith("v_rcp_f32") for x in gen) >= 8 * stores // 3
        assert sum(x.startswith("v_cmp_lt_f32") for x in gen) >= 8 * stores // 3
        # no lane exchange in the generated text (board_row_channel order): up to its far jump to the seam
        gen_end = gen.index(next(x for x in gen if x.startswith("s_setpc_b64")))
        assert not [x for x in gen[:gen_end] if x.startswith("v_permlane16_swap")], "the generated epilogue exchanges lanes"
        assert sum(x.startswith("s_setpc_b64") for x in tail) >= 1, "the generated epilogue jumps to the seam"
        # the seam: every exit goes through  vmcnt(0) -> s_barrier -> (end | next element)
        seam = sections[f"tower{w}_seam"]
        ends = [i for i, x in enumerate(seam) if x.startswith("s_endpgm")]
        assert len(ends) == 1 and not [x for x in tail + head if x.startswith("s_endpgm")], "the seam owns the only s_endpgm"
        last = seam[:ends[0]]
        assert any(x.startswith("s_waitcnt vmcnt(0)") for x in last) and any(x.startswith("s_barrier") for x in last), last
        assert any(x.startswith("s_setpc_b64") for x in last), "far jump back to the dispatch"
        # the FC body: no AGPR, no MFMA, VGPRs only from the hook's range, leaves through far jumps (no s_endpgm)
        fc = sections[f"tower{w}_body_fc"]
        assert len(fc) > 200 and not [x for x in fc if x.startswith("s_endpgm") or "v_accvgpr" in x or x.startswith("v_mfma")]
        regs = [int(n) for x in fc for n in re.findall(r"\bv(\d+)\b", x)] + [int(n) for x in fc for pair in re.findall(r"\bv\[(\d+):(\d+)\]", x) for n in pair]
        assert regs and min(regs) >= 66 and max(regs) <= 127, (min(regs), max(regs))
        assert sum(x.startswith("s_setpc_b64") for x in fc) >= 1
    notes = subprocess.run([READELF, "--notes", hsaco], capture_output=True, text=True, check=True).stdout
    blocks = [b for b in notes.split(".agpr_count") if "conv_tower_kernelILi4EEE" in b]
    assert blocks and re.search(r"\.group_segment_fixed_size:\s+163840", blocks[0]), "launch kernel: static 160 KiB of LDS"
    assert re.search(r"\.private_segment_fixed_size:\s+0\b", blocks[0]), "launch kernel: no scratch"
    assert re.search(r"\.vgpr_spill_count:\s+0\b", blocks[0]) and re.search(r"\.sgpr_spill_count:\s+0\b", blocks[0])


def test_seam_rejection_builds_without_the_tower(tmp_path, capfd):
    """tower_seam.py depends on the register assignment hipcc produced (validated on ROCm 7.2).  When it does not recognise the
    compiler's assembly -- here: an instruction planted between the accumulator anchors and the hook, and an anchor that names a
    register inside the hook's clobber range -- the BUILD must go on without the persistent kernel: an empty blob
    (`sayuri_tower_hsaco_size` = 0) is linked, engine.hip's load_tower_module reports it and the per-layer kernels run.  The
    unperturbed assembly must still go through."""
    asm = os.path.join(_build.LIB, "obj", "tower.s")
    if not os.path.exists(asm):
        _build.build_tower_blob(force=True)
    text = open(asm).read()
    hook = text.index("; TOWER_SE_HOOK ")
    line0 = text.rfind("\n", 0, hook) + 1
    planted = text[:line0] + "\tv_mov_b32_e32 v1, v2\n" + text[line0:]
    anchor = re.search(r"; TOWER_ACC 0 0 0 a\[(\d+):(\d+)\]", text)
    assert anchor, "no anchor of tile (0, 0) in tower.s"
    moved = text.replace(anchor.group(0), "; TOWER_ACC 0 0 0 v[100:103]", 1)
    probe = tmp_path / "probe.c"
    probe.write_text('#include <stdio.h>\nextern const unsigned char sayuri_tower_hsaco[];\nextern const unsigned long long sayuri_tower_hsaco_size;\n'
                     'int main(void) { printf("%llu %d\\n", sayuri_tower_hsaco_size, (int)sayuri_tower_hsaco[1]); return 0; }\n')

    def blob_size(objdir):
        exe = os.path.join(objdir, "probe")
        subprocess.check_call(["gcc", str(probe), os.path.join(objdir, "tower_blob.o"), "-o", exe])
        size, second = subprocess.check_output([exe], text=True).split()
        return int(size), int(second)

    os.environ.pop("SAYURI_TOWER_REQUIRED", None)
    for name, body in (("planted", planted), ("moved", moved)):
        d = tmp_path / name
        d.mkdir()
        (d / "tower.s").write_text(body)
        blob, ok = _build.tower_blob_from_asm(str(d / "tower.s"), str(d))
        err = capfd.readouterr().err
        assert not ok and os.path.exists(blob), name
        assert "WITHOUT the persistent tower kernel" in err and "tower_seam.py rejected" in err, err
        assert blob_size(str(d)) == (0, 0)
        os.environ["SAYURI_TOWER_REQUIRED"] = "1"
        try:
            with pytest.raises(RuntimeError):
                _build.tower_blob_from_asm(str(d / "tower.s"), str(d))
        finally:
            del os.environ["SAYURI_TOWER_REQUIRED"]
    good = tmp_path / "good"
    good.mkdir()
    (good / "tower.s").write_text(text)
    blob, ok = _build.tower_blob_from_asm(str(good / "tower.s"), str(good))
    size, second = blob_size(str(good))
    assert ok and size > 100000 and second == ord("E")   # "\x7fELF"
    # ... and the shipped library carries the real thing
    assert os.path.getsize(os.path.join(_build.LIB, "obj", "tower.hsaco")) == size
