#!/usr/bin/env python3
"""Close the layer loop of the persistent tower launch in the assembly hipcc emits for tower.hip, and put the SE unit
into it as generated assembly.

    python tower_seam.py tower.s tower_seamed.s [--inv] [--sleep=N] [--align=A --pad=P]

conv_tower.h explains why neither is written in C++.  hipcc compiles, per channel-tile width W, two single-layer kernels
    conv_tower_kernel<W>    K loop | hook | epilogue        (the convolution body)
    tower_se_fc_kernel<W>   pooled partials in LDS -> gate in LDS   (the FCs of the SE unit)
each with ONE argument, a pointer to a TowerLayer, loaded from s[0:1] + 0.  This script turns the pair into one persistent
kernel (entered through the conv_tower_kernel<W> symbol):

  entry stub     s[B:B+1] <- the table pointer (the launch's only argument), s[B+2] <- workgroup id, s[B+3] <- wave id,
                 where B is the first SGPR no compiled body allocates
  dispatch       rebuilds the ABI entry state for the element at s[B:B+1] -- s[0:1] = its address (the element starts
                 with its own address, so it reads as a kernarg segment), s2 = workgroup id, v0 = thread id, exec = -1 --
                 parks the element's has_se in s[B+4] and enters the convolution body
  hook           the `; TOWER_SE_HOOK` asm statement behind the K loop becomes: has_se == 0 ? nothing :
                     pooling over the accumulators IN THE REGISTERS THE K LOOP LEFT THEM IN (read off the `; TOWER_ACC`
                     anchors that precede the hook) -> partial sums / maxima in LDS, the FC images on their way by LDS-DMA
                     far jump into the FC body (its VGPRs renamed into the range the hook statement clobbers), which
                     returns through every one of its s_endpgm
                     the gate, x <- sigmoid(gamma) x + beta, in place
  seam           every s_endpgm of the convolution body becomes a branch to:  s_waitcnt vmcnt(0) lgkmcnt(0) (this wave's
                 stores are acknowledged), s_barrier (all eight waves'), then end if the element was the last of the run, else
                 s[B:B+1] += STRIDE and back to dispatch.

The launch kernel's descriptor and metadata get the union of the bodies' resources (SGPRs incl. the five parked ones) and the
whole 160 KiB of LDS as its static group segment (the bodies address it from 0).  Scratch: none -- the build fails if any body
wants it.  Fails loudly when the assembly does not look like what it was written against.

What the generated pooling / gate must reproduce bit for bit is board_se_pool / board_se_gate (conv_board.h): the per-layer
kernel conv_board_se_kernel runs those, and SAYURI_TOWER=0/1 give identical outputs (tests/test_gpu_smallops.py).
"""
import re
import sys

STRIDE = 320       # sizeof(TowerLayer), conv_tower.h kTowerStride
LAST_OFFSET = 8    # offsetof(TowerLayer, last); has_se follows it
LDS_BYTES = 160 * 1024
FREE_VGPR = 66     # conv_tower.h kTowerFreeVgpr: v[66:127] belong to the hook
FREE_SGPRS = 64    # conv_tower.h kTowerFreeSgprs: s[0:63] belong to the hook
FC_SGPR_CAP = 40   # the FC body may use s[0:39]; the hook keeps what must survive the call in s[40:63]
NJ = 12            # conv_board.h kBoardNJ

CONV_RE = re.compile(r"^(_ZN6sayuri17conv_tower_kernelILi(\d+)EEEvPKNS_10TowerLayerE):")
FC_RE = re.compile(r"^(_ZN6sayuri18tower_se_fc_kernelILi(\d+)EEEvPKNS_10TowerLayerE):")


def die(msg):
    sys.stderr.write("tower_seam.py: " + msg + "\n")
    sys.exit(1)


def far_jump(sym, t=4):
    # the idiom LLVM emits for a far call; s[t:t+1] is dead at every place this script uses it
    return [f"\ts_getpc_b64 s[{t}:{t + 1}]",
            f"\ts_add_u32 s{t}, s{t}, {sym}@rel32@lo+4",
            f"\ts_addc_u32 s{t + 1}, s{t + 1}, {sym}@rel32@hi+12",
            f"\ts_setpc_b64 s[{t}:{t + 1}]"]


def is_instruction(ln):
    s = ln.strip()
    return bool(s) and ln.startswith("\t") and not s.startswith(".") and not s.startswith(";")


def rename_vgprs(ln, shift):
    """v<N> -> v<N+shift> in the operands of one line of compiled assembly (mnemonics never match: `v_`)."""
    code, sep, comment = ln.partition(";")
    code = re.sub(r"\bv\[(\d+):(\d+)\]", lambda m: f"v[{int(m.group(1)) + shift}:{int(m.group(2)) + shift}]", code)
    code = re.sub(r"\bv(\d+)\b", lambda m: f"v{int(m.group(1)) + shift}", code)
    return code + sep + comment


def max_reg(body, cls):
    hi = -1
    for ln in body:
        code = ln.split(";")[0]
        for m in re.finditer(r"\b%s\[(\d+):(\d+)\]" % cls, code):
            hi = max(hi, int(m.group(2)))
        for m in re.finditer(r"\b%s(\d+)\b" % cls, code):
            hi = max(hi, int(m.group(1)))
    return hi


class Regs:
    """Names for the hook's registers: v[FREE_VGPR:127] and s[0:FREE_SGPRS-1]; `keep` SGPRs survive the FC call."""

    def __init__(self):
        self.v_next, self.s_low, self.s_keep = FREE_VGPR, 4, FC_SGPR_CAP  # s[0:3] are set up for the call itself

    def v(self, n=1, align=1):
        self.v_next = (self.v_next + align - 1) // align * align
        r = self.v_next
        self.v_next += n
        if self.v_next > 128:
            die("the hook ran out of VGPRs")
        return r

    def s(self, n=1, keep=False):
        if keep:
            self.s_keep = (self.s_keep + n - 1) // n * n
            r = self.s_keep
            self.s_keep += n
            if self.s_keep > FREE_SGPRS:
                die("the hook ran out of SGPRs")
            return r
        self.s_low = (self.s_low + n - 1) // n * n
        r = self.s_low
        self.s_low += n
        if self.s_low > FC_SGPR_CAP:
            die("the hook ran out of scratch SGPRs")
        return r


def quad_channel(i):
    """conv_board.h board_quad_channel for even tile counts, lane quad 0: first channel (from the wave's first) of the four a lane
    holds of row tile i; quad q adds 8 q"""
    return (i >> 1) * 32 + 4 * (i & 1)


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def sr(lo, n=1):
    return f"s{lo}" if n == 1 else f"s[{lo}:{lo + n - 1}]"


def se_hook(w, hook, acc, B, fc_label):
    """The SE unit between the K loop and the epilogue of width-w's convolution body, as assembly text.
    hook: the parsed operands of the TOWER_SE_HOOK statement; acc[(i, j)] = ('a' | 'v', first register) of output tile
    (row tile i, column tile j); B: first parked SGPR (table B:B+1, workgroup B+2, wave B+3, has_se B+4)."""
    wmt, kot = hook["wmt"], hook["kot"]
    if wmt % 2:
        die("the SE hook is written for even row-tile counts (board_row_channel order)")
    E, T = hook["elem"], hook["tid"]          # s[lo:lo+1] text, v text
    WG, WAVE, HAS_SE = f"s{B + 2}", f"s{B + 3}", f"s{B + 4}"
    L = f".Ltower{w}_se"
    R = Regs()
    o = []
    a = o.append

    # ---- registers
    s_m0 = R.s(keep=True)
    s_w1h, s_w2h = R.s(2, keep=True), R.s(2, keep=True)
    s_wb = R.s(2, keep=True)                  # w1_bytes, w2_bytes
    s_wbytes = R.s(keep=True)                 # bytes of LDS in front of the stage's vectors
    s_wave_m = R.s(keep=True)
    s_info, s_ncols, s_bs, s_nj0, s_wave_n, s_col0, s_nj, s_njm1 = (R.s() for _ in range(8))
    s_t0, s_t1, s_k, s_n = (R.s() for _ in range(4))
    s_p = R.s(2)
    s_valid, s_px0, s_save = R.s(2), R.s(2), R.s(2)
    v_lane, v_lane16, v_px, v_q, v_t, v_addr, v_addr2 = (R.v() for _ in range(7))
    s_ro = R.s(keep=True)                     # BoardParams::row_order of the layer
    sets = [(R.v(4, 4), R.v(4, 4)) for _ in range(2)]     # (sums, maxima) of a row tile, alternating
    xs = [R.v(4, 4) for _ in range(2)]                    # an AGPR tile on its way through the VALU, alternating

    a(f"\t; ---- tower_seam.py: the SE unit (width {w}: {wmt} row tiles x {NJ} column tiles per wave)")
    a(f"\ts_cmp_eq_u32 {HAS_SE}, 0")
    a(f"\ts_cbranch_scc1 {L}_skip")
    a(f"\ts_mov_b32 {sr(s_m0)}, m0")
    a("\ts_waitcnt lgkmcnt(0)")
    a("\ts_barrier")                          # every wave is done with the rings
    a(f"\ts_load_dwordx2 {sr(s_w1h, 2)}, {E}, {hex(hook['w1h'])}")
    a(f"\ts_load_dwordx2 {sr(s_w2h, 2)}, {E}, {hex(hook['w2h'])}")
    if hook["w2b"] != hook["w1b"] + 4:
        die("w1_bytes / w2_bytes are not adjacent in BoardSeParams")
    a(f"\ts_load_dwordx2 {sr(s_wb, 2)}, {E}, {hex(hook['w1b'])}")
    a(f"\ts_load_dword {sr(s_info)}, {E}, {hex(hook['ui'])}")
    a(f"\tv_and_b32_e32 {vr(v_lane)}, 63, {T}")
    a(f"\tv_lshlrev_b32_e32 {vr(v_lane16)}, 4, {vr(v_lane)}")
    a(f"\tv_and_b32_e32 {vr(v_px)}, 15, {vr(v_lane)}")
    a(f"\tv_lshrrev_b32_e32 {vr(v_q)}, 4, {vr(v_lane)}")
    a("\ts_waitcnt lgkmcnt(0)")
    a(f"\ts_cmp_gt_i32 {sr(s_info)}, -1")
    a(f"\ts_cbranch_scc1 {L}_info")
    a(f"\ts_load_dwordx2 {sr(s_p, 2)}, {E}, {hex(hook['cols'])}")
    a(f"\ts_lshl_b32 {sr(s_t0)}, {WG}, 2")
    a("\ts_waitcnt lgkmcnt(0)")
    a(f"\ts_load_dword {sr(s_info)}, {sr(s_p, 2)}, {sr(s_t0)}")
    a("\ts_waitcnt lgkmcnt(0)")
    a(f"{L}_info:")
    a(f"\ts_and_b32 {sr(s_ncols)}, {sr(s_info)}, 0xff")
    a(f"\ts_lshr_b32 {sr(s_bs)}, {sr(s_info)}, 8")
    a(f"\ts_add_u32 {sr(s_nj0)}, {sr(s_ncols)}, 1")
    a(f"\ts_lshr_b32 {sr(s_nj0)}, {sr(s_nj0)}, 1")
    a(f"\ts_lshr_b32 {sr(s_wave_n)}, {WAVE}, 2")
    a(f"\ts_and_b32 {sr(s_wave_m)}, {WAVE}, 3")
    a(f"\ts_sub_u32 {sr(s_t0)}, {sr(s_ncols)}, {sr(s_nj0)}")
    a(f"\ts_cmp_eq_u32 {sr(s_wave_n)}, 0")
    a(f"\ts_cselect_b32 {sr(s_col0)}, 0, {sr(s_nj0)}")
    a(f"\ts_cselect_b32 {sr(s_nj)}, {sr(s_nj0)}, {sr(s_t0)}")
    a(f"\ts_sub_u32 {sr(s_njm1)}, {sr(s_nj)}, 1")
    a(f"\ts_add_u32 {sr(s_t0)}, {sr(s_wb)}, {sr(s_wb + 1)}")
    a(f"\ts_cmp_lg_u64 {sr(s_w1h, 2)}, 0")
    a(f"\ts_cselect_b32 {sr(s_wbytes)}, {sr(s_t0)}, 0")
    a(f"\ts_cbranch_scc0 {L}_nostage")
    # ---- both FC images on their way into LDS: linear 1 KiB pieces, piece k by wave k % 8; the squeeze image of this
    # tile's board size first
    a(f"\ts_sub_u32 {sr(s_t0)}, {sr(s_bs)}, 2")
    a(f"\ts_mul_hi_u32 {sr(s_t1)}, {sr(s_t0)}, {sr(s_wb)}")
    a(f"\ts_mul_i32 {sr(s_t0)}, {sr(s_t0)}, {sr(s_wb)}")
    a(f"\ts_add_u32 {sr(s_w1h)}, {sr(s_w1h)}, {sr(s_t0)}")
    a(f"\ts_addc_u32 {sr(s_w1h + 1)}, {sr(s_w1h + 1)}, {sr(s_t1)}")
    for tag, base, nbytes, lds0 in (("1", s_w1h, s_wb, None), ("2", s_w2h, s_wb + 1, s_wb)):
        a(f"\ts_lshr_b32 {sr(s_n)}, {sr(nbytes)}, 10")
        a(f"\ts_mov_b32 {sr(s_k)}, {WAVE}")
        a(f"{L}_dma{tag}:")
        a(f"\ts_cmp_ge_u32 {sr(s_k)}, {sr(s_n)}")
        a(f"\ts_cbranch_scc1 {L}_dma{tag}_done")
        a(f"\ts_lshl_b32 {sr(s_t0)}, {sr(s_k)}, 10")
        a(f"\ts_add_u32 {sr(s_p)}, {sr(base)}, {sr(s_t0)}")
        a(f"\ts_addc_u32 {sr(s_p + 1)}, {sr(base + 1)}, 0")
        if lds0 is not None:
            a(f"\ts_add_u32 {sr(s_t0)}, {sr(s_t0)}, {sr(lds0)}")
        a(f"\ts_mov_b32 m0, {sr(s_t0)}")
        a("\ts_nop 4")
        a(f"\tglobal_load_lds_dwordx4 {vr(v_lane16)}, {sr(s_p, 2)}")
        a(f"\ts_add_u32 {sr(s_k)}, {sr(s_k)}, 8")
        a(f"\ts_branch {L}_dma{tag}")
        a(f"{L}_dma{tag}_done:")
    a(f"{L}_nostage:")
    # ---- pooling.  last_valid (lane): (col0 + nj - 1) * 16 + px < bs * bs -- a one-sample tile has its unused pixel slots at
    # the end, only the wave's last column tile can hold any
    a(f"\ts_add_u32 {sr(s_t0)}, {sr(s_col0)}, {sr(s_njm1)}")
    a(f"\ts_lshl_b32 {sr(s_t0)}, {sr(s_t0)}, 4")
    a(f"\tv_add_u32_e32 {vr(v_t)}, {sr(s_t0)}, {vr(v_px)}")
    a(f"\ts_mul_i32 {sr(s_t1)}, {sr(s_bs)}, {sr(s_bs)}")
    a(f"\tv_cmp_gt_u32_e64 {sr(s_valid, 2)}, {sr(s_t1)}, {vr(v_t)}")
    a(f"\tv_cmp_eq_u32_e64 {sr(s_px0, 2)}, 0, {vr(v_px)}")
    # psum[wave_n * KO_T + wave_m * WMT * 16 + first channel of (row tile i, quad q)] (floats) behind the images
    a(f"\ts_mul_i32 {sr(s_t0)}, {sr(s_wave_n)}, {kot * 4}")
    a(f"\ts_mul_i32 {sr(s_t1)}, {sr(s_wave_m)}, {wmt * 16 * 4}")
    a(f"\ts_add_u32 {sr(s_t0)}, {sr(s_t0)}, {sr(s_t1)}")
    a(f"\ts_add_u32 {sr(s_t0)}, {sr(s_t0)}, {sr(s_wbytes)}")
    a(f"\ts_add_u32 {sr(s_t0)}, {sr(s_t0)}, {hook['psum']}")
    a(f"\tv_lshl_add_u32 {vr(v_addr)}, {vr(v_q)}, 4, {sr(s_t0)}")       # natural row order: quad q of row tile i at 16 i + 4 q
    a(f"\tv_lshl_add_u32 {vr(v_addr2)}, {vr(v_q)}, 5, {sr(s_t0)}")      # board_row_channel order: at quad_channel(i) + 8 q
    a(f"\ts_load_dword {sr(s_ro)}, {E}, {hex(hook['roword'])}")

    def tile_ops(i, j, S4, M4, X):
        """sum += tile, max = max(max, tile) for output tile (i, j); the caller has set exec"""
        kind, lo = acc[(i, j)]
        if kind == "a":
            for r in range(4):
                a(f"\tv_accvgpr_read_b32 {vr(X + r)}, a{lo + r}")
            src = X
        else:
            src = lo
        if src % 2 == 0:
            a(f"\tv_pk_add_f32 {vr(S4, 2)}, {vr(S4, 2)}, {vr(src, 2)}")
            a(f"\tv_pk_add_f32 {vr(S4 + 2, 2)}, {vr(S4 + 2, 2)}, {vr(src + 2, 2)}")
        else:
            for r in range(4):
                a(f"\tv_add_f32_e32 {vr(S4 + r)}, {vr(S4 + r)}, {vr(src + r)}")
        for r in range(4):
            a(f"\tv_max_f32_e32 {vr(M4 + r)}, {vr(M4 + r)}, {vr(src + r)}")

    for i in range(wmt):
        S4, M4 = sets[i & 1]
        for r in range(4):
            a(f"\tv_mov_b32_e32 {vr(S4 + r)}, 0")
            a(f"\tv_mov_b32_e32 {vr(M4 + r)}, 0xc59c4000")   # -5000.0f
        a(f"\ts_cmp_lt_i32 {sr(s_nj)}, 1")
        a(f"\ts_cbranch_scc1 {L}_red{i}")
        for j in range(NJ):
            a(f"\ts_cmp_eq_u32 {sr(s_njm1)}, {j}")
            a(f"\ts_cbranch_scc1 {L}_last{i}_{j}")
            tile_ops(i, j, S4, M4, xs[j & 1])
        a(f"\ts_branch {L}_red{i}")
        for j in range(NJ):
            a(f"{L}_last{i}_{j}:")
            a(f"\ts_mov_b64 {sr(s_save, 2)}, exec")
            a(f"\ts_and_b64 exec, exec, {sr(s_valid, 2)}")
            tile_ops(i, j, S4, M4, xs[j & 1])
            a(f"\ts_mov_b64 exec, {sr(s_save, 2)}")
            if j + 1 < NJ:
                a(f"\ts_branch {L}_red{i}")
        a(f"{L}_red{i}:")
        a("\ts_nop 4")   # a VALU result read by a DPP operand: two wait states; exec written by the SALU: five
        for step in (8, 4, 2, 1):
            for r in range(4):
                a(f"\tv_add_f32_dpp {vr(S4 + r)}, {vr(S4 + r)}, {vr(S4 + r)} row_ror:{step} row_mask:0xf bank_mask:0xf")
            for r in range(4):
                a(f"\tv_max_f32_dpp {vr(M4 + r)}, {vr(M4 + r)}, {vr(M4 + r)} row_ror:{step} row_mask:0xf bank_mask:0xf")
        a(f"\ts_mov_b64 {sr(s_save, 2)}, exec")
        a(f"\ts_and_b64 exec, exec, {sr(s_px0, 2)}")
        if i == 0:
            a("\ts_waitcnt lgkmcnt(0)")      # row_order has arrived
        a(f"\ts_cmp_eq_u32 {sr(s_ro)}, 0")
        a(f"\ts_cbranch_scc1 {L}_pst{i}")
        a(f"\tds_write_b128 {vr(v_addr2)}, {vr(S4, 4)} offset:{quad_channel(i) * 4}")
        a(f"\tds_write_b128 {vr(v_addr2)}, {vr(M4, 4)} offset:{quad_channel(i) * 4 + hook['pmax'] - hook['psum']}")
        a(f"\ts_branch {L}_pse{i}")
        a(f"{L}_pst{i}:")
        a(f"\tds_write_b128 {vr(v_addr)}, {vr(S4, 4)} offset:{i * 64}")
        a(f"\tds_write_b128 {vr(v_addr)}, {vr(M4, 4)} offset:{i * 64 + hook['pmax'] - hook['psum']}")
        a(f"{L}_pse{i}:")
        a(f"\ts_mov_b64 exec, {sr(s_save, 2)}")
    a("\ts_waitcnt vmcnt(0) lgkmcnt(0)")    # this wave's pieces of the images have landed, its partials are written
    a("\ts_barrier")
    # ---- the FCs: a compiled body of its own, entered with the ABI's entry state
    e_lo = int(re.match(r"s\[(\d+):", E).group(1))
    a(f"\ts_mov_b32 s0, s{e_lo}")
    a(f"\ts_mov_b32 s1, s{e_lo + 1}")
    a(f"\ts_mov_b32 s2, {WG}")
    a(f"\tv_mov_b32_e32 v{FREE_VGPR}, {T}")
    a("\ts_mov_b64 exec, -1")
    o.extend(far_jump(fc_label))
    a(f"tower{w}_se_return:")
    a("\ts_mov_b64 exec, -1")
    # ---- the gate: x <- sigmoid(gamma) x + beta, one fused multiply-add per value, in place
    R2 = Regs()
    g_lane, g_q, g_addr = R2.v(), R2.v(), R2.v()
    gs = [(R2.v(4, 4), R2.v(4, 4)) for _ in range(wmt)]
    gx = [R2.v(4, 4) for _ in range(2)]
    a(f"\tv_and_b32_e32 {vr(g_lane)}, 63, {T}")
    a(f"\tv_lshrrev_b32_e32 {vr(g_q)}, 4, {vr(g_lane)}")
    a(f"\ts_mul_i32 s4, {sr(s_wave_m)}, {wmt * 16 * 4}")
    a(f"\ts_add_u32 s4, s4, {sr(s_wbytes)}")
    a(f"\ts_add_u32 s4, s4, {hook['gate']}")
    a(f"\ts_cmp_eq_u32 {sr(s_ro)}, 0")
    a(f"\ts_cbranch_scc1 {L}_gnat")
    a(f"\tv_lshl_add_u32 {vr(g_addr)}, {vr(g_q)}, 5, s4")
    for i in range(wmt):
        a(f"\tds_read_b128 {vr(gs[i][0], 4)}, {vr(g_addr)} offset:{quad_channel(i) * 4}")
        a(f"\tds_read_b128 {vr(gs[i][1], 4)}, {vr(g_addr)} offset:{quad_channel(i) * 4 + kot * 4}")
    a(f"\ts_branch {L}_gread")
    a(f"{L}_gnat:")
    a(f"\tv_lshl_add_u32 {vr(g_addr)}, {vr(g_q)}, 4, s4")
    for i in range(wmt):
        a(f"\tds_read_b128 {vr(gs[i][0], 4)}, {vr(g_addr)} offset:{i * 64}")
        a(f"\tds_read_b128 {vr(gs[i][1], 4)}, {vr(g_addr)} offset:{i * 64 + kot * 4}")
    a(f"{L}_gread:")
    a("\ts_waitcnt lgkmcnt(0)")
    a("\ts_barrier")                         # the gate is read: the epilogue's residual rows may land in this LDS

    def fma_tile(G, Bt, t):
        if t % 2 == 0:
            a(f"\tv_pk_fma_f32 {vr(t, 2)}, {vr(G, 2)}, {vr(t, 2)}, {vr(Bt, 2)}")
            a(f"\tv_pk_fma_f32 {vr(t + 2, 2)}, {vr(G + 2, 2)}, {vr(t + 2, 2)}, {vr(Bt + 2, 2)}")
        else:
            for r in range(4):
                a(f"\tv_fma_f32 {vr(t + r)}, {vr(G + r)}, {vr(t + r)}, {vr(Bt + r)}")

    tiles = [(i, j) for i in range(wmt) for j in range(NJ)]
    agpr = [t for t in tiles if acc[t][0] == "a"]
    for k in range(0, len(agpr), 2):
        pair = agpr[k:k + 2]
        for n, t in enumerate(pair):
            for r in range(4):
                a(f"\tv_accvgpr_read_b32 {vr(gx[n] + r)}, a{acc[t][1] + r}")
        for n, t in enumerate(pair):
            fma_tile(gs[t[0]][0], gs[t[0]][1], gx[n])
        for n, t in enumerate(pair):
            for r in range(4):
                a(f"\tv_accvgpr_write_b32 a{acc[t][1] + r}, {vr(gx[n] + r)}")
    for t in tiles:
        if acc[t][0] == "v":
            fma_tile(gs[t[0]][0], gs[t[0]][1], acc[t][1])
    a(f"\ts_mov_b32 m0, {sr(s_m0)}")
    a(f"{L}_skip:")
    return o


def epi_hook(w, hook, acc, B):
    """The epilogue of width-w's convolution body as assembly text, for the layers the host marks with row_order = 1 (Mish, ReLU or
    no activation, one sample per tile with computed table entries, the layer's channels = the channel tile; their weights and bias are in
    board_row_channel order, so a lane's two accumulator quads of a row-tile pair ARE 8 consecutive channels): optional residual, activation, fp16 NHWC store, straight
    from the accumulators where the K loop left them -- the arithmetic of board_epilogue / mish2 (conv_board.h) operation for
    operation, so the outputs equal the compiled epilogue's bit for bit.  What it saves is what hipcc adds: per 16-byte store
    ~14 register moves staging its operands and a conversion + half a packed add per residual value (here one
    v_fma_mix_f32), in front of everything ~200 accumulator moves -- the epilogue is bound by VALU issue (tools/ubench/trans_rate.hip:
    a plain operation 4 cycles of a SIMD, a packed one 5, a transcendental 10.7, v_permlane16_swap 14).  Other layers fall through
    to the compiled epilogue behind this text."""
    wmt, kot = hook["wmt"], hook["kot"]
    if wmt % 2:
        return []
    npair = wmt // 2
    pieces = NJ * npair
    nd = max(pieces - 20, 0)            # residual pieces that go to registers instead of the wave's 20 KiB of LDS
    if nd % npair:
        die("epilogue: register pieces are not whole column tiles")
    JH, NRT = NJ // 2, nd // npair      # register tiles: [JH, JH + NRT)
    E, T = hook["elem"], hook["tid"]
    WG, WAVE = f"s{B + 2}", f"s{B + 3}"
    L = f".Ltower{w}_ep"
    R = Regs()
    o = []
    a = o.append
    s_res, s_out, s_l2e, s_valid, s_save = R.s(2), R.s(2), R.s(2), R.s(2), R.s(2)
    (s_arith, s_act, s_couts, s_slotpix, s_ui, s_ncols, s_bs, s_nj0, s_wave_n, s_wave_m, s_col0, s_nj, s_npix, s_t0, s_t1, s_late,
     s_step, s_mylds) = (R.s() for _ in range(18))
    v_lane, v_R, v_px, v_cb, v_off, v_pxl, v_t, v_lds, v_a, v_r, v_so = (R.v() for _ in range(11))
    rreg = [R.v(4, 4) for _ in range(nd)]
    X, Y, RR = R.v(4, 4), R.v(4, 4), R.v(4, 4)
    Tm, U = R.v(8, 2), R.v(8, 4)
    Hs = [U, U + 4]   # the fp16 result of a pair lives where the pair's temporaries lived (alternating halves: a store's data is
                      # rewritten two pairs later at the earliest)

    def lds_slot(j, pr):
        return (j if j < JH else j - NRT) * npair + pr

    a(f"\t; ---- tower_seam.py: the epilogue (width {w}) for Mish layers with computed table entries; others take the compiled one below")
    a(f"\ts_load_dword {sr(s_arith)}, {E}, {hex(hook['roword'])}")
    a(f"\ts_load_dword {sr(s_act)}, {E}, {hex(hook['act'])}")
    a(f"\ts_load_dword {sr(s_couts)}, {E}, {hex(hook['couts'])}")
    a(f"\ts_load_dword {sr(s_slotpix)}, {E}, {hex(hook['slotpix'])}")
    a(f"\ts_load_dword {sr(s_ui)}, {E}, {hex(hook['ui'])}")
    a(f"\ts_load_dwordx2 {sr(s_res, 2)}, {E}, {hex(hook['res'])}")
    a(f"\ts_load_dwordx2 {sr(s_out, 2)}, {E}, {hex(hook['out'])}")
    a("\ts_waitcnt lgkmcnt(0)")
    # row_order = 1: the host gave this layer the board_row_channel image BECAUSE this text will run (Mish, one sample per tile with
    # computed table entries, channels = the channel tile: Engine::board_row_order_ok)
    a(f"\ts_cmp_eq_u32 {sr(s_arith)}, 0")
    a(f"\ts_cbranch_scc1 {L}_compiled")
    # row_order = 3 (SAYURI_TOWER_NOEPI_AFTER=n, a MEASURING switch of the engine): this layer's epilogue is skipped -- nothing is
    # stored, the activations stay what the last complete forward left (realistic operands for the next layer's MFMAs, unlike a
    # build that never stores: that one multiplies zeros and gains clock).  What a launch costs without its epilogues bounds
    # what hiding them under the MFMA stream could be worth.
    a(f"\ts_cmp_eq_u32 {sr(s_arith)}, 3")
    a(f"\ts_cbranch_scc1 {L}_done")
    # ---- geometry of this wave and lane (board_epilogue's first lines)
    a(f"\tv_and_b32_e32 {vr(v_lane)}, 63, {T}")
    a(f"\tv_lshrrev_b32_e32 {vr(v_R)}, 4, {vr(v_lane)}")
    a(f"\tv_and_b32_e32 {vr(v_px)}, 15, {vr(v_lane)}")
    a(f"\ts_and_b32 {sr(s_ncols)}, {sr(s_ui)}, 0xff")
    a(f"\ts_lshr_b32 {sr(s_bs)}, {sr(s_ui)}, 8")
    a(f"\ts_add_u32 {sr(s_nj0)}, {sr(s_ncols)}, 1")
    a(f"\ts_lshr_b32 {sr(s_nj0)}, {sr(s_nj0)}, 1")
    a(f"\ts_lshr_b32 {sr(s_wave_n)}, {WAVE}, 2")
    a(f"\ts_and_b32 {sr(s_wave_m)}, {WAVE}, 3")
    a(f"\ts_sub_u32 {sr(s_t0)}, {sr(s_ncols)}, {sr(s_nj0)}")
    a(f"\ts_cmp_eq_u32 {sr(s_wave_n)}, 0")
    a(f"\ts_cselect_b32 {sr(s_col0)}, 0, {sr(s_nj0)}")
    a(f"\ts_cselect_b32 {sr(s_nj)}, {sr(s_nj0)}, {sr(s_t0)}")
    a(f"\ts_mul_i32 {sr(s_npix)}, {sr(s_bs)}, {sr(s_bs)}")
    # the weight image's rows are in board_row_channel order: a lane holds channels 32 pr + 8 R .. + 7 of pair pr (the quad of
    # row tile 2 pr, then the quad of 2 pr + 1) -- cb(pr) = wave_m * WMT * 16 + 8 R + 32 pr, no exchange between lanes
    a(f"\ts_mul_i32 {sr(s_t0)}, {sr(s_wave_m)}, {wmt * 16}")
    a(f"\tv_lshl_add_u32 {vr(v_cb)}, {vr(v_R)}, 3, {sr(s_t0)}")
    # byte offset of (this lane's pixel of column tile 0, cb(0)) in the output / residual buffers; column tile j adds j * step
    a(f"\ts_lshl_b32 {sr(s_t1)}, {sr(s_col0)}, 4")
    a(f"\tv_add_u32_e32 {vr(v_pxl)}, {sr(s_t1)}, {vr(v_px)}")
    a(f"\ts_mul_i32 {sr(s_t0)}, {WG}, {sr(s_slotpix)}")
    a(f"\tv_add_u32_e32 {vr(v_off)}, {sr(s_t0)}, {vr(v_pxl)}")
    a(f"\tv_mul_lo_u32 {vr(v_off)}, {vr(v_off)}, {sr(s_couts)}")
    a(f"\tv_add_u32_e32 {vr(v_off)}, {vr(v_off)}, {vr(v_cb)}")
    a(f"\tv_lshlrev_b32_e32 {vr(v_off)}, 1, {vr(v_off)}")
    a(f"\ts_lshl_b32 {sr(s_step)}, {sr(s_couts)}, 5")
    a(f"\ts_mul_i32 {sr(s_mylds)}, {WAVE}, {20 * 1024}")
    a(f"\tv_lshlrev_b32_e32 {vr(v_lds)}, 4, {vr(v_lane)}")
    a(f"\tv_add_u32_e32 {vr(v_lds)}, {sr(s_mylds)}, {vr(v_lds)}")
    a(f"\ts_mov_b32 {sr(s_l2e)}, 0x3fb8aa3b")
    a(f"\ts_mov_b32 {sr(s_l2e + 1)}, 0x3fb8aa3b")
    a(f"\ts_cmp_eq_u64 {sr(s_res, 2)}, 0")
    a(f"\ts_cbranch_scc1 {L}_tiles_nores")
    # ---- residual: ALL rows of the wave requested at once (LDS-DMA into the dead rings, the few that do not fit into
    # registers); issue order = first half, second half, register pieces: board_epilogue explains the two-step wait
    a("\ts_waitcnt lgkmcnt(0)")
    a("\ts_barrier")                     # every wave is done with the rings

    def offsets_of(j, want_pr):
        """v_a <- byte offset of (column tile j, pair 0), vcc <- lanes whose pixel exists; yields the piece offsets in v_r"""
        a(f"\tv_add_u32_e32 {vr(v_t)}, {16 * j}, {vr(v_pxl)}")
        a(f"\tv_cmp_gt_u32_e32 vcc, {sr(s_npix)}, {vr(v_t)}")
        a(f"\ts_mul_i32 {sr(s_t0)}, {sr(s_step)}, {j}")
        a(f"\tv_add_u32_e32 {vr(v_a)}, {sr(s_t0)}, {vr(v_off)}")

    lds_tiles = [j for j in range(NJ) if j < JH or j >= JH + NRT]
    for j in lds_tiles:
        a(f"\ts_cmp_le_u32 {sr(s_nj)}, {j}")
        a(f"\ts_cbranch_scc1 {L}_lds_issued")
        offsets_of(j, None)
        for pr in range(npair):
            if pr:
                a(f"\tv_add_u32_e32 {vr(v_a)}, 64, {vr(v_a)}")
            a(f"\tv_cndmask_b32_e32 {vr(v_r)}, 0, {vr(v_a)}, vcc")
            a(f"\ts_add_u32 {sr(s_t1)}, {sr(s_mylds)}, {lds_slot(j, pr) * 1024}")
            a(f"\ts_mov_b32 m0, {sr(s_t1)}")
            a("\ts_nop 1")
            a(f"\tglobal_load_lds_dwordx4 {vr(v_r)}, {sr(s_res, 2)}")
    a(f"{L}_lds_issued:")
    for k in range(nd):
        j, pr = JH + k // npair, k % npair
        if pr == 0:
            a(f"\ts_cmp_le_u32 {sr(s_nj)}, {j}")
            a(f"\ts_cbranch_scc1 {L}_reg_issued")
            offsets_of(j, None)
        else:
            a(f"\tv_add_u32_e32 {vr(v_a)}, 64, {vr(v_a)}")
        a(f"\tv_cndmask_b32_e32 {vr(v_r)}, 0, {vr(v_a)}, vcc")
        a(f"\tglobal_load_dwordx4 {vr(rreg[k], 4)}, {vr(v_r)}, {sr(s_res, 2)}")
    a(f"{L}_reg_issued:")
    # loads younger than the first half = the pieces of the column tiles [JH, nj)
    a(f"\ts_sub_i32 {sr(s_late)}, {sr(s_nj)}, {JH}")
    a(f"\ts_max_i32 {sr(s_late)}, {sr(s_late)}, 0")
    for t in range(NJ - JH + 1):
        a(f"\ts_cmp_eq_u32 {sr(s_late)}, {t}")
        a(f"\ts_cbranch_scc1 {L}_late{t}")
    a(f"\ts_branch {L}_late0")
    for t in range(NJ - JH, -1, -1):
        a(f"{L}_late{t}:")
        a(f"\ts_waitcnt vmcnt({t * npair})")
        a(f"\ts_branch {L}_tiles")

    def mish8(x_pairs, h):
        """x_pairs: four even-aligned VGPR pairs holding 8 values; h <- their Mish as 8 fp16 (4 VGPRs)"""
        t = [Tm + 2 * k for k in range(4)]
        u = [U + 2 * k for k in range(4)]
        for k in range(4):
            a(f"\tv_pk_mul_f32 {vr(t[k], 2)}, {vr(x_pairs[k], 2)}, {sr(s_l2e, 2)}")
        for k in range(8):
            a(f"\tv_exp_f32_e32 {vr(Tm + k)}, {vr(Tm + k)}")
        for k in range(4):
            a(f"\tv_pk_add_f32 {vr(u[k], 2)}, {vr(t[k], 2)}, 2.0 op_sel_hi:[1,0]")
        for k in range(4):
            a(f"\tv_pk_fma_f32 {vr(t[k], 2)}, {vr(t[k], 2)}, {vr(u[k], 2)}, 2.0 op_sel_hi:[1,1,0]")
        for k in range(8):
            a(f"\tv_rcp_f32_e32 {vr(Tm + k)}, {vr(Tm + k)}")
        for k in range(4):
            a(f"\tv_pk_fma_f32 {vr(t[k], 2)}, {vr(t[k], 2)}, -2.0, 1.0 op_sel_hi:[1,0,0]")
        for k in range(4):
            a(f"\tv_pk_mul_f32 {vr(x_pairs[k], 2)}, {vr(x_pairs[k], 2)}, {vr(t[k], 2)}")
        for k in range(4):
            a(f"\tv_cvt_pk_f16_f32 {vr(h + k)}, {vr(x_pairs[k])}, {vr(x_pairs[k] + 1)}")

    def act8(kind, x_pairs, h):
        """h <- act(the 8 values of x_pairs) as 8 fp16; kind = the activation's name in common.h"""
        if kind == "mish":
            return mish8(x_pairs, h)
        if kind == "relu":      # x > 0 ? x : 0 as a compare and a select (v_max_f32 might hand back -0)
            for k in range(4):
                for e in range(2):
                    a(f"\tv_cmp_lt_f32_e32 vcc, 0, {vr(x_pairs[k] + e)}")
                    a(f"\tv_cndmask_b32_e32 {vr(x_pairs[k] + e)}, 0, {vr(x_pairs[k] + e)}, vcc")
        for k in range(4):      # (identity: nothing but the conversion)
            a(f"\tv_cvt_pk_f16_f32 {vr(h + k)}, {vr(x_pairs[k])}, {vr(x_pairs[k] + 1)}")

    def tiles(with_res, kind):
        tag = ("r" if with_res else "n") + kind
        nstore = 0
        for j in range(NJ):
            a(f"\ts_cmp_le_u32 {sr(s_nj)}, {j}")
            a(f"\ts_cbranch_scc1 {L}_done")
            if j == JH and with_res:
                a("\ts_waitcnt vmcnt(0)")         # the second half's rows (and the first half's stores)
            a(f"\tv_add_u32_e32 {vr(v_t)}, {16 * j}, {vr(v_pxl)}")
            a(f"\tv_cmp_gt_u32_e64 {sr(s_valid, 2)}, {sr(s_npix)}, {vr(v_t)}")
            a(f"\ts_mul_i32 {sr(s_t0)}, {sr(s_step)}, {j}")
            a(f"\tv_add_u32_e32 {vr(v_so)}, {sr(s_t0)}, {vr(v_off)}")
            for pr in range(npair):
                ta, tb = acc[(2 * pr, j)], acc[(2 * pr + 1, j)]
                in_lds = not (JH <= j < JH + NRT)
                rr = RR
                if with_res and in_lds:
                    # the piece comes back from LDS (lane-linear: the lane that reads it is the lane that fetched it)
                    a(f"\tds_read_b128 {vr(RR, 4)}, {vr(v_lds)} offset:{lds_slot(j, pr) * 1024}")
                elif with_res:
                    rr = rreg[(j - JH) * npair + pr]
                xs = []
                for (cls, lo), tmp in ((ta, X), (tb, Y)):
                    if cls == "a":
                        for r in range(4):
                            a(f"\tv_accvgpr_read_b32 {vr(tmp + r)}, a{lo + r}")
                        xs.append(tmp)
                    else:
                        if lo % 2:
                            die("epilogue: an accumulator tile in VGPRs is not even-aligned")
                        xs.append(lo)
                xa, xb = xs
                if with_res:
                    # + residual: one fused multiply-add per value reads the fp16 half directly (v + (float)rr * 1.0: the same
                    # single rounding as conversion + add)
                    if in_lds:
                        a("\ts_waitcnt lgkmcnt(0)")
                    vals = [xa, xa + 1, xa + 2, xa + 3, xb, xb + 1, xb + 2, xb + 3]
                    for q in range(8):
                        a(f"\tv_fma_mix_f32 {vr(vals[q])}, {vr(rr + q // 2)}, 1.0, {vr(vals[q])} op_sel:[{q & 1},0,0] op_sel_hi:[1,0,0]")
                h = Hs[nstore & 1]
                nstore += 1
                act8(kind, [xa, xa + 2, xb, xb + 2], h)
                a(f"\ts_and_saveexec_b64 {sr(s_save, 2)}, {sr(s_valid, 2)}")
                a(f"\tglobal_store_dwordx4 {vr(v_so)}, {vr(h, 4)}, {sr(s_out, 2)}" + (f" offset:{64 * pr}" if pr else ""))
                a(f"\ts_mov_b64 exec, {sr(s_save, 2)}")
        a(f"\ts_branch {L}_done")

    # one pair of tile loops (with / without residual) per activation the text covers
    kinds = (("mish", hook["mish"]), ("relu", hook["relu"]), ("identity", hook["identity"]))
    for with_res in (True, False):
        a(f"{L}_tiles:" if with_res else f"{L}_tiles_nores:")
        for kind, code in kinds[1:]:
            a(f"\ts_cmp_eq_u32 {sr(s_act)}, {code}")
            a(f"\ts_cbranch_scc1 {L}_{'r' if with_res else 'n'}{kind}")
        for kind, code in kinds:
            if kind != "mish":
                a(f"{L}_{'r' if with_res else 'n'}{kind}:")
            tiles(with_res, kind)
    a(f"{L}_done:")
    o.extend(far_jump(f"tower{w}_seam"))
    a(f"{L}_compiled:")
    return o


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    inv = "--inv" in sys.argv
    # measuring builds only: --sleep=N parks every wave for N x 64 cycles in the seam (what does a stall cost a chip that
    # runs at its power limit?)
    sleep = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--sleep=")), 0)
    # placement of the convolution body: it starts on a 2^align-byte boundary plus pad bytes (control never falls into a body,
    # so the padding is never executed); the K loop sits at a fixed distance from the body's first instruction
    align = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--align=")), 0)
    pad = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--pad=")), 0)
    place = ([f"\t.p2align {align}"] if align else []) + (["\ts_nop 0"] * (pad // 4))
    if len(args) != 2:
        die("usage: tower_seam.py in.s out.s [--inv]")
    lines = open(args[0]).read().split("\n")

    # ---- locate the bodies: label line .. the line before `.section .rodata` that follows the last s_endpgm
    funcs = {}
    i = 0
    while i < len(lines):
        m = CONV_RE.match(lines[i]) or FC_RE.match(lines[i])
        if m:
            kind = "conv" if CONV_RE.match(lines[i]) else "fc"
            name, w = m.group(1), int(m.group(2))
            j = i + 1
            while j < len(lines) and not lines[j].startswith("\t.section\t.rodata"):
                j += 1
            if j == len(lines):
                die("no .rodata section after " + name)
            funcs[(w, kind)] = dict(name=name, begin=i, end=j)
            i = j
        i += 1
    widths = sorted({w for (w, _) in funcs})
    if not widths:
        die("no conv_tower_kernel in the input")
    for w in widths:
        if (w, "conv") not in funcs or (w, "fc") not in funcs:
            die(f"conv_tower_kernel<{w}> and tower_se_fc_kernel<{w}> must both be present")

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
        conv, fc = funcs[(w, "conv")], funcs[(w, "fc")]
        for f in (conv, fc):
            for key, want in (("user_sgpr_count", 2), ("user_sgpr_kernarg_segment_ptr", 1), ("system_sgpr_workgroup_id_x", 1),
                              ("system_sgpr_workgroup_id_y", 0), ("system_sgpr_workgroup_id_z", 0), ("system_vgpr_workitem_id", 0),
                              ("kernarg_size", 8), ("group_segment_fixed_size", 0), ("user_sgpr_kernarg_preload_length", 0),
                              ("uses_dynamic_stack", 0), ("private_segment_fixed_size", 0)):
                got, _ = directive(f["name"], key)
                if got != want:
                    die(f"{f['name']}: .amdhsa_{key} = {got}, the seam was written for {want}")
            body = lines[f["begin"] + 1:f["end"]]
            first = next((ln for ln in body if is_instruction(ln)), "")
            if not re.match(r"\ts_load_dwordx2 s\[\d+:\d+\], s\[0:1\], 0x0", first):
                die(f"{f['name']}: the body does not start by loading its argument from s[0:1] ({first.strip()!r})")
            if any(re.match(r"\s*scratch_", ln) for ln in body):
                die(f"{f['name']}: scratch access in a tower body")
            f["sgprs"], _ = directive(f["name"], "next_free_sgpr")
            f["vgprs"], _ = directive(f["name"], "next_free_vgpr")
            f["accum"], _ = directive(f["name"], "accum_offset")

        # ---- the FC body: a subroutine of the hook.  No AGPRs, its VGPRs fit behind FREE_VGPR, its SGPRs below FC_SGPR_CAP
        fbody = lines[fc["begin"] + 1:fc["end"]]
        if any(re.search(r"\ba\d+\b|\ba\[\d+:\d+\]|v_accvgpr|v_mfma", ln.split(";")[0]) for ln in fbody):
            die(f"{fc['name']}: touches AGPRs (they hold the accumulators)")
        fv, fs = max_reg(fbody, "v"), max_reg(fbody, "s")
        if fv + FREE_VGPR > 127:
            die(f"{fc['name']}: v{fv} does not fit behind v{FREE_VGPR} (128 - {FREE_VGPR} VGPRs belong to the hook)")
        if fs >= FC_SGPR_CAP:
            die(f"{fc['name']}: s{fs} -- the hook keeps its own values from s{FC_SGPR_CAP} on")
        if any(re.search(r"\bm0\b|ttmp|flat_scratch|s_getpc|s_setpc|s_swappc|s_call", ln.split(";")[0]) for ln in fbody):
            die(f"{fc['name']}: uses m0 / calls (not expected in the FC body)")
        fc_label, fc_ret = f"tower{w}_body_fc", f"tower{w}_se_return"
        for k in range(fc["begin"] + 1, fc["end"]):
            if lines[k].strip() == "s_endpgm":
                edits[k] = far_jump(fc_ret)
            elif is_instruction(lines[k]):
                edits[k] = [rename_vgprs(lines[k], FREE_VGPR)]
        edits[fc["begin"]] = [lines[fc["begin"]], f"{fc_label}:", "\t; ---- compiled body (FCs of the SE unit), VGPRs renamed by tower_seam.py"]

        # ---- the convolution body: anchors and hook
        cb = range(conv["begin"] + 1, conv["end"])
        hooks = [k for k in cb if "; TOWER_SE_HOOK " in lines[k]]
        if len(hooks) != 1:
            die(f"{conv['name']}: {len(hooks)} TOWER_SE_HOOK statements")
        hk = hooks[0]
        hook = {}
        for tok in lines[hk].split("TOWER_SE_HOOK", 1)[1].split():
            key, _, val = tok.partition("=")
            hook[key] = val if key in ("elem", "tid") else int(val, 0)
        for key in ("elem", "tid", "wmt", "ui", "cols", "w1h", "w2h", "w1b", "w2b", "psum", "pmax", "gate", "kot", "res", "out", "couts",
                    "slotpix", "act", "arith", "mish", "relu", "identity", "roword"):
            if key not in hook:
                die(f"{conv['name']}: the hook statement names no `{key}`")
        if hook["wmt"] != w:
            die(f"{conv['name']}: the hook says wmt={hook['wmt']}")
        m = re.match(r"^s\[(\d+):(\d+)\]$", hook["elem"])
        if not m or int(m.group(1)) < FREE_SGPRS or not re.match(r"^v\d+$", hook["tid"]) or int(hook["tid"][1:]) >= FREE_VGPR:
            die(f"{conv['name']}: hook operands {hook['elem']} / {hook['tid']} sit inside the clobbered ranges")
        acc, first_anchor = {}, None
        for k in cb:
            m = re.search(r"; TOWER_ACC (\d+) (\d+) (\d+) ([av])\[(\d+):(\d+)\]", lines[k])
            if not m:
                continue
            side, ti, tj, kind, lo, hi = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5)), int(m.group(6))
            if side != 0 or hi != lo + 3 or (ti, tj) in acc or k > hk:
                die(f"{conv['name']}: unexpected anchor {lines[k].strip()!r}")
            if kind == "v" and lo + 3 >= FREE_VGPR:
                die(f"{conv['name']}: accumulator tile ({ti}, {tj}) lives in v[{lo}:{hi}], inside the hook's range")
            acc[(ti, tj)] = (kind, lo)
            first_anchor = k if first_anchor is None else first_anchor
        if len(acc) != w * NJ:
            die(f"{conv['name']}: {len(acc)} anchors for {w * NJ} output tiles")
        regs = sorted((kind, lo) for kind, lo in acc.values())
        if any(regs[n][0] == regs[n + 1][0] and regs[n][1] + 4 > regs[n + 1][1] for n in range(len(regs) - 1)):
            die(f"{conv['name']}: overlapping accumulator tiles")
        stray = [lines[k].strip() for k in range(first_anchor, hk) if is_instruction(lines[k])]
        if stray:
            die(f"{conv['name']}: instructions between the anchors and the hook ({stray[:3]}): the tiles may have moved")
        # the MFMAs' results are read by the pooling: the K loop's s_nop 15 pair must be what precedes the anchors
        before = [lines[k].strip() for k in range(conv["begin"] + 1, first_anchor) if is_instruction(lines[k])][-2:]
        if before != ["s_nop 15", "s_nop 15"]:
            die(f"{conv['name']}: {before} in front of the anchors, expected the K loop's two s_nop 15")
        # every accumulator the MFMA stream writes is an anchored tile
        written = set()
        for k in range(conv["begin"] + 1, first_anchor):
            m = re.match(r"\tv_mfma_\w+ ([av])\[(\d+):(\d+)\]", lines[k])
            if m:
                written.add((m.group(1), int(m.group(2))))
        if written != set(acc.values()):
            die(f"{conv['name']}: the MFMA stream writes {len(written)} tiles, the anchors name {len(set(acc.values()))} (or others)")

        B = (max(conv["sgprs"], fc["sgprs"]) + 1) & ~1
        if B + 5 > 102:
            die(f"width {w}: no five SGPRs left above the compiler's {B}")
        if B < FREE_SGPRS + 2:
            B = FREE_SGPRS + 2
        tab, wg, wv, hs = f"s[{B}:{B + 1}]", f"s{B + 2}", f"s{B + 3}", f"s{B + 4}"
        disp, body_conv = f"tower{w}_dispatch", f"tower{w}_body_conv"
        edits[hk] = se_hook(w, hook, acc, B, fc_label) + epi_hook(w, hook, acc, B)

        entry = [f"\t; ---- tower_seam.py: entry stub (table {tab}, workgroup {wg}, wave {wv}, has_se {hs})",
                 f"\ts_load_dwordx2 {tab}, s[0:1], 0x0",
                 f"\ts_mov_b32 {wg}, s2",
                 f"\tv_readfirstlane_b32 {wv}, v0",
                 "\ts_nop 0",
                 f"\ts_lshr_b32 {wv}, {wv}, 6",
                 "\ts_waitcnt lgkmcnt(0)",
                 f"{disp}:",
                 f"\ts_load_dword {hs}, {tab}, {hex(LAST_OFFSET + 4)}",
                 "\ts_mov_b64 exec, -1",
                 f"\ts_mov_b32 s0, s{B}",
                 f"\ts_mov_b32 s1, s{B + 1}",
                 f"\ts_mov_b32 s2, {wg}",
                 "\tv_mbcnt_lo_u32_b32 v0, -1, 0",
                 "\tv_mbcnt_hi_u32_b32 v0, -1, v0",
                 f"\tv_lshl_or_b32 v0, {wv}, 6, v0",
                 "\ts_waitcnt lgkmcnt(0)",
                 f"\ts_branch {body_conv}"] + place + [
                 f"{body_conv}:",
                 "\t; ---- compiled body (convolution: K loop, hook, epilogue)"]
        edits[conv["begin"]] = [lines[conv["begin"]]] + entry

        seam, done = f".Ltower{w}_seam", f".Ltower{w}_done"
        ends = [k for k in cb if lines[k].strip() == "s_endpgm"]
        if not ends:
            die(f"{conv['name']}: no s_endpgm")
        for k in ends:
            edits[k] = [f"\ts_branch {seam}"]
        tail = [f"{seam}:",
                f"tower{w}_seam:",
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
        if conv["accum"] < FREE_VGPR + fv + 1:
            die(f"width {w}: accum_offset {conv['accum']} below the FC body's renamed VGPRs")
        for key, val in (("next_free_sgpr", B + 5), ("group_segment_fixed_size", LDS_BYTES)):
            _, idx = directive(conv["name"], key)
            edits[idx] = [re.sub(r"\d+\s*$", str(val), lines[idx])]
        # ---- and its metadata entry (the runtime sizes LDS from there)
        try:
            n = next(k for k, ln in enumerate(lines) if ln.strip().startswith(".name:") and ln.split()[-1] == conv["name"])
        except StopIteration:
            die("no metadata entry for " + conv["name"])
        lo = n
        while not lines[lo].startswith("  - "):
            lo -= 1
        hi = n
        while hi + 1 < len(lines) and not lines[hi + 1].startswith("  - ") and not lines[hi + 1].startswith("amdhsa.") and lines[hi + 1].startswith("    "):
            hi += 1
        seen = set()
        for k in range(lo, hi + 1):
            for key, val in ((".group_segment_fixed_size:", LDS_BYTES), (".sgpr_count:", B + 5 + 6)):
                if lines[k].strip().lstrip("- ").startswith(key):
                    edits[k] = [re.sub(r"\d+\s*$", str(val), lines[k])]
                    seen.add(key)
        if len(seen) != 2:
            die("metadata entry of " + conv["name"] + " lacks " + str(2 - len(seen)) + " expected keys")

    out = []
    for k, ln in enumerate(lines):
        out.extend(edits.get(k, [ln]))
    # one .text section for everything: the seam's far jumps then resolve at assembly time and no comdat group is cut
    text = "\n".join(out)
    text = re.sub(r'^\t\.section\t\.text\.[^\n]*,comdat$', "\t.text", text, flags=re.M)
    open(args[1], "w").write(text)


if __name__ == "__main__":
    main()
