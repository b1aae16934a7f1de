#!/usr/bin/env python3
"""Close the layer loop of the persistent tower launch in the assembly hipcc emits for tower.hip.

    python tower_seam.py tower.s tower_seamed.s [--inv]

conv_tower.h explains why the loop is not written in C++.  hipcc compiles, per channel-tile width W, two single-layer
kernels  conv_tower_kernel<W, false>  (plain convolution)  and  conv_tower_kernel<W, true>  (convolution + SE unit);
each takes ONE argument, a pointer to a TowerLayer, loaded from s[0:1] + 0.  This script turns the pair into one
persistent kernel (entered through the <W, false> symbol):

  entry stub     s[B:B+1] <- the table pointer (the launch's only argument), s[B+2] <- workgroup id, s[B+3] <- wave id,
                 where B is the first SGPR neither compiled body allocates
  dispatch       rebuilds the ABI entry state for the element at s[B:B+1] -- s[0:1] = its address (the element starts
                 with its own address, so it reads as a kernarg segment), s2 = workgroup id, v0 = thread id, exec = -1 --
                 and enters the body the element's has_se asks for
  seam           every s_endpgm of both bodies becomes a branch to:  s_waitcnt vmcnt(0) lgkmcnt(0) (this wave's stores
                 are acknowledged), s_barrier (all eight waves'), then end if the element was the last of the run, else
                 s[B:B+1] += STRIDE and back to dispatch.

The launch kernel's descriptor and metadata get the union of both bodies' resources (SGPRs incl. the four parked ones,
scratch) and the whole 160 KiB of LDS as its static group segment (the bodies address it from 0).
Fails loudly when the assembly does not look like what it was written against.
"""
import re
import sys

STRIDE = 320       # sizeof(TowerLayer), conv_tower.h kTowerStride
LAST_OFFSET = 8    # offsetof(TowerLayer, last); has_se follows it
LDS_BYTES = 160 * 1024

KERNEL_RE = re.compile(r"^(_ZN6sayuri17conv_tower_kernelILi(\d+)ELb([01])EEEvPKNS_10TowerLayerE):")


def die(msg):
    sys.stderr.write("tower_seam.py: " + msg + "\n")
    sys.exit(1)


def far_jump(sym, t=4):
    # the idiom LLVM emits for a far call; s[t:t+1] is dead at every place this script uses it
    return [f"\ts_getpc_b64 s[{t}:{t + 1}]",
            f"\ts_add_u32 s{t}, s{t}, {sym}@rel32@lo+4",
            f"\ts_addc_u32 s{t + 1}, s{t + 1}, {sym}@rel32@hi+12",
            f"\ts_setpc_b64 s[{t}:{t + 1}]"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    inv = "--inv" in sys.argv
    # measuring builds only: --sleep=N parks every wave for N x 64 cycles in the seam (what does a stall cost a chip that
    # runs at its power limit?)
    sleep = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--sleep=")), 0)
    # placement of the compiled bodies: each starts on a 2^align-byte boundary plus pad bytes (control never falls into a body,
    # so the padding is never executed); the K loop sits at a fixed distance from the body's first instruction
    align = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--align=")), 0)
    pad = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--pad=")), 0)
    place = ([f"\t.p2align {align}"] if align else []) + (["\ts_nop 0"] * (pad // 4))
    pad_se = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--pad-se=")), 0)  # the SE body: its symbol is 256-aligned
    place_se = ["\ts_nop 0"] * (pad_se // 4)
    if len(args) != 2:
        die("usage: tower_seam.py in.s out.s [--inv]")
    lines = open(args[0]).read().split("\n")

    # ---- locate the bodies: label line .. the line before `.section .rodata` that follows the last s_endpgm
    funcs = {}
    i = 0
    while i < len(lines):
        m = KERNEL_RE.match(lines[i])
        if m:
            name, w, se = m.group(1), int(m.group(2)), int(m.group(3))
            j = i + 1
            while j < len(lines) and not lines[j].startswith("\t.section\t.rodata"):
                j += 1
            if j == len(lines):
                die("no .rodata section after " + name)
            funcs[(w, se)] = dict(name=name, begin=i, end=j)
            i = j
        i += 1
    widths = sorted({w for (w, _) in funcs})
    if not widths:
        die("no conv_tower_kernel in the input")
    for w in widths:
        if (w, 0) not in funcs or (w, 1) not in funcs:
            die(f"conv_tower_kernel<{w}, false/true> must both be present")

    def directive(name, key):
        """value of `.amdhsa_<key>` inside the descriptor of kernel `name` (and its line index)"""
        k = None
        for idx, ln in enumerate(lines):
            if ln.strip() == ".amdhsa_kernel " + name:
                k = idx
            elif k is not None and ln.strip().startswith(".amdhsa_" + key + " "):
                return int(ln.split()[-1]), idx
            elif k is not None and ln.strip() == ".end_amdhsa_kernel":
                break
        die(f"descriptor of {name}: no .amdhsa_{key}")

    edits = {}      # line index -> replacement list
    for w in widths:
        plain, se = funcs[(w, 0)], funcs[(w, 1)]
        for f in (plain, se):
            for key, want in (("user_sgpr_count", 2), ("user_sgpr_kernarg_segment_ptr", 1), ("system_sgpr_workgroup_id_x", 1),
                              ("system_sgpr_workgroup_id_y", 0), ("system_sgpr_workgroup_id_z", 0), ("system_vgpr_workitem_id", 0),
                              ("kernarg_size", 8), ("group_segment_fixed_size", 0), ("user_sgpr_kernarg_preload_length", 0),
                              ("uses_dynamic_stack", 0)):
                got, _ = directive(f["name"], key)
                if got != want:
                    die(f"{f['name']}: .amdhsa_{key} = {got}, the seam was written for {want}")
            body = lines[f["begin"] + 1:f["end"]]
            first = next((ln for ln in body if ln.startswith("\t") and not ln.startswith("\t.")), "")
            if not re.match(r"\ts_load_dwordx2 s\[\d+:\d+\], s\[0:1\], 0x0", first):
                die(f"{f['name']}: the body does not start by loading its argument from s[0:1] ({first.strip()!r})")
            f["sgprs"], _ = directive(f["name"], "next_free_sgpr")
            f["scratch"], _ = directive(f["name"], "private_segment_fixed_size")
            f["vgprs"], _ = directive(f["name"], "next_free_vgpr")
            f["accum"], _ = directive(f["name"], "accum_offset")
        if plain["accum"] != se["accum"]:
            die(f"width {w}: the two bodies put their AGPRs at different offsets of the register file")
        vgprs = max(plain["vgprs"], se["vgprs"])  # same accum_offset: the launch needs the larger of the two totals
        B = (max(plain["sgprs"], se["sgprs"]) + 1) & ~1
        if B + 4 > 102:
            die(f"width {w}: no four SGPRs left above the compiler's {B}")
        tab, wg, wv = f"s[{B}:{B + 1}]", f"s{B + 2}", f"s{B + 3}"
        disp, body_plain, body_se = f"tower{w}_dispatch", f"tower{w}_body_plain", f"tower{w}_body_se"

        entry = [f"\t; ---- tower_seam.py: entry stub (table {tab}, workgroup {wg}, wave {wv})",
                 f"\ts_load_dwordx2 {tab}, s[0:1], 0x0",
                 f"\ts_mov_b32 {wg}, s2",
                 f"\tv_readfirstlane_b32 {wv}, v0",
                 "\ts_nop 0",
                 f"\ts_lshr_b32 {wv}, {wv}, 6",
                 "\ts_waitcnt lgkmcnt(0)",
                 f"{disp}:",
                 f"\ts_load_dwordx2 s[4:5], {tab}, {hex(LAST_OFFSET)}",
                 "\ts_mov_b64 exec, -1",
                 f"\ts_mov_b32 s0, s{B}",
                 f"\ts_mov_b32 s1, s{B + 1}",
                 f"\ts_mov_b32 s2, {wg}",
                 "\tv_mbcnt_lo_u32_b32 v0, -1, 0",
                 "\tv_mbcnt_hi_u32_b32 v0, -1, v0",
                 f"\tv_lshl_or_b32 v0, {wv}, 6, v0",
                 "\ts_waitcnt lgkmcnt(0)",
                 "\ts_cmp_eq_u32 s5, 0",
                 f"\ts_cbranch_scc1 {body_plain}"] + far_jump(body_se) + place + [
                 f"{body_plain}:",
                 "\t; ---- compiled body (plain convolution)"]
        edits[plain["begin"]] = [lines[plain["begin"]]] + entry
        edits[se["begin"]] = [lines[se["begin"]]] + place_se + [f"{body_se}:", "\t; ---- compiled body (convolution + SE unit)"]

        for tag, f in (("p", plain), ("s", se)):
            seam, done = f".Ltower{w}{tag}_seam", f".Ltower{w}{tag}_done"
            ends = [k for k in range(f["begin"], f["end"]) if lines[k].strip() == "s_endpgm"]
            if not ends:
                die(f"{f['name']}: no s_endpgm")
            for k in ends:
                edits[k] = [f"\ts_branch {seam}"]
            tail = [f"{seam}:",
                    "\ts_waitcnt vmcnt(0) lgkmcnt(0)",
                    f"\ts_load_dword s4, {tab}, {hex(LAST_OFFSET)}",
                    "\ts_waitcnt lgkmcnt(0)",
                    "\ts_barrier"]
            if inv:
                tail.append("\tbuffer_inv sc1")
            for _ in range(sleep // 127):
                tail.append("\ts_sleep 127")
            if sleep % 127:
                tail.append(f"\ts_sleep {sleep % 127}")
            tail += ["\ts_cmp_lg_u32 s4, 0",
                     f"\ts_cbranch_scc1 {done}",
                     f"\ts_add_u32 s{B}, s{B}, {STRIDE}",
                     f"\ts_addc_u32 s{B + 1}, s{B + 1}, 0"] + far_jump(disp) + [
                     f"{done}:",
                     "\ts_endpgm"]
            edits[ends[-1]] = edits[ends[-1]] + tail

        # ---- the launch kernel's descriptor
        scratch = max(plain["scratch"], se["scratch"])
        for key, val in (("next_free_vgpr", vgprs), ("next_free_sgpr", B + 4), ("private_segment_fixed_size", scratch), ("enable_private_segment", 1 if scratch else 0),
                         ("group_segment_fixed_size", LDS_BYTES)):
            _, idx = directive(plain["name"], key)
            edits[idx] = [re.sub(r"\d+\s*$", str(val), lines[idx])]
        # ---- and its metadata entry (the runtime sizes LDS and scratch from there)
        try:
            n = next(k for k, ln in enumerate(lines) if ln.strip() == ".name:           " + plain["name"] or
                     (ln.strip().startswith(".name:") and ln.split()[-1] == plain["name"]))
        except StopIteration:
            die("no metadata entry for " + plain["name"])
        lo = n
        while not lines[lo].startswith("  - "):
            lo -= 1
        hi = n
        while hi + 1 < len(lines) and not lines[hi + 1].startswith("  - ") and not lines[hi + 1].startswith("amdhsa.") and lines[hi + 1].startswith("    "):
            hi += 1
        seen = set()
        for k in range(lo, hi + 1):
            for key, val in ((".group_segment_fixed_size:", LDS_BYTES), (".private_segment_fixed_size:", scratch), (".sgpr_count:", B + 4 + 6), (".vgpr_count:", vgprs),
                             (".agpr_count:", vgprs - plain["accum"])):
                if lines[k].strip().lstrip("- ").startswith(key):
                    edits[k] = [re.sub(r"\d+\s*$", str(val), lines[k])]
                    seen.add(key)
        if len(seen) != 5:
            die("metadata entry of " + plain["name"] + " lacks " + str(5 - len(seen)) + " expected keys")

    out = []
    for k, ln in enumerate(lines):
        out.extend(edits.get(k, [ln]))
    # one .text section for everything: the seam's far jumps then resolve at assembly time and no comdat group is cut
    text = "\n".join(out)
    text = re.sub(r'^\t\.section\t\.text\.[^\n]*,comdat$', "\t.text", text, flags=re.M)
    open(args[1], "w").write(text)


if __name__ == "__main__":
    main()
