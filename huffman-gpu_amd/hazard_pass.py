#!/usr/bin/env python3
"""Build driver of libhvlc.so with one pass of its own over the device assembly.

Why.  gfx950 code that reads an SGPR as (part of) a memory address too soon after a VALU instruction wrote it can see
the OLD value.  The ISA asks for 5 wait states between the two and the compiler inserts exactly those (`s_nop 3` in

    v_readlane_b32 s62, v79, 2        ; reload of a spilled pointer
    v_readlane_b32 s63, v79, 3
    v_readlane_b32 s61, v79, 1
    s_nop 3
    global_load_dwordx2 v[14:15], v2, s[62:63] sc1

), yet on MI355X this load used, about once in 10^8 executions, the stale high half (fault address = {new low, old high})
and once both stale halves (6 wait states were not enough either); with 16 more wait states patched into the same
binary it never did (round 2, profiles/README.md).  k_encode lives at the SGPR limit, so where the register allocator
puts such reloads changes with every edit.  Instead of chasing them in the source, this pass makes the sequence
impossible: in front of every memory instruction that reads an SGPR a VALU instruction may have written within the last
`window` wait states ON ANY PATH, it inserts

    s_mov_b32 sN, sN

A SALU read of a VALU-written SGPR is interlocked in hardware (every v_readfirstlane -> s_add relies on it), and a
SALU-written SGPR may feed a memory instruction right away: the copy onto itself is a precise wait for the register.

The root cause, found with tools/hazard_repro.hip (profiles/r02_hazard_repro.txt): an `s_wakeup` executed by ANOTHER
wave of the workgroup ends an `s_nop N` in progress the way it ends an `s_sleep` -- the wait states the compiler counted
on are silently cut short (quiet neighbours: 0 stale reads in 10^9; neighbours sending s_wakeup: 0.5 % stale; N + 1
separate `s_nop 0`, or the SALU copies: 0 again).  k_encode's waves wake each other.  So, second measure, in every kernel
that contains an s_wakeup the pass splits each `s_nop N` into N + 1 `s_nop 0` -- every instruction takes its own issue
slot whatever wakes the wave -- which restores ALL the compiler's software-managed hazards (DPP, lane select, ...), not
only the one that was seen to bite.

How.  hipcc -S (device only) -> control-flow graph per kernel from the labels and branches -> forward data flow of
"wait states since a VALU instruction last wrote sN" (minimum over the paths) to a fixed point -> the copies -> assemble,
link, bundle, and compile the host side around the device binary (-fcuda-include-gpubinary).  build_check.py then scans
the finished library (straight-line, on the disassembly) as an independent check.

    python huffman-gpu_amd/hazard_pass.py -o libhvlc.so csrc/hvlc_api.hip [-DNAME[=VALUE] ...]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def llvm_bin():
    """Directory of clang / ld.lld / clang-offload-bundler / llvm-objdump of the ROCm installation hipcc belongs to."""
    import shutil
    cands = []
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "clang")):
            return c
    raise RuntimeError("hazard pass: no ROCm LLVM tools found (looked in %s)" % ", ".join(cands))


LLVM = llvm_bin()
ARCH = "gfx950"
GUARD = "-DHVLC_HAZARD_PASS"          # hvlc_api.hip refuses to compile without it: only this script may build the library
WINDOW = 48                       # wait states below which a VALU-written SGPR is not trusted in an address

SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]|\b(vcc)\b|\b(vcc_lo)\b|\b(vcc_hi)\b")
VMEM_PREFIX = ("global_", "flat_", "scratch_", "buffer_", "tbuffer_")
SMEM_PREFIX = ("s_load_", "s_buffer_load_", "s_store_", "s_atomic_", "s_scratch_", "s_dcache_")
CARRY_OUT = ("_co_", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")
# (s_cmov*: no write when SCC = 0, so the register may still hold the VALU value; it stays "young".  s_store_* / s_atomic_*
# without return / s_dcache_*: the first operand is data or nothing, not a destination.)
NO_SDST = ("s_cmov_b32", "s_cmov_b64", "s_cmovk_i32",
           "s_nop", "s_waitcnt", "s_sleep", "s_branch", "s_barrier", "s_endpgm", "s_wakeup", "s_setprio", "s_sethalt",
           "s_sendmsg", "s_trap", "s_icache_inv", "s_incperflevel", "s_decperflevel", "s_ttracedata", "s_setreg_b32",
           "s_setreg_imm32_b32", "s_setvskip", "s_code_end", "s_set_gpr_idx_on", "s_set_gpr_idx_off", "s_set_gpr_idx_mode")
NREG = 108                        # s0..s105 (+ vcc as 106, 107)


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        elif m.group(2):
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.update((106, 107))
        elif m.group(5):
            out.add(106)
        elif m.group(6):
            out.add(107)
    return {r for r in out if r < NREG}


def reg_name(r):
    return "vcc_lo" if r == 106 else "vcc_hi" if r == 107 else "s%d" % r


def split_operands(ops):
    return [p.strip() for p in ops.split(",")] if ops.strip() else []


class Ins:
    __slots__ = ("line", "mnem", "ops", "reads_addr", "valu_writes", "salu_writes", "ws", "target", "kind")

    def __init__(self, line, mnem, ops):
        self.line, self.mnem, self.ops = line, mnem, ops
        parts = split_operands(ops)
        self.reads_addr, self.valu_writes, self.salu_writes = set(), set(), set()
        self.ws = 1 + (int(ops.strip() or "0", 0) if mnem == "s_nop" else 0)
        self.target, self.kind = None, "plain"
        if mnem.startswith(VMEM_PREFIX):
            self.reads_addr = sregs(ops)
        elif mnem.startswith(SMEM_PREFIX):
            if parts and mnem.startswith(("s_load_", "s_buffer_load_", "s_scratch_load_")):
                self.salu_writes = sregs(parts[0])           # (the destination; SMEM addresses are interlocked like SALU reads)
        elif mnem.startswith("v_"):
            if parts:
                self.valu_writes = sregs(parts[0])
                if len(parts) > 1 and any(k in mnem for k in CARRY_OUT):
                    self.valu_writes |= sregs(parts[1])
            if mnem.startswith("v_cmp") and "_e64" not in mnem and "_sdwa" not in mnem and not mnem.startswith("v_cmpx"):
                self.valu_writes |= {106, 107}
        elif mnem.startswith("s_"):
            if mnem == "s_branch":
                self.kind, self.target = "jump", ops.strip()
            elif mnem.startswith("s_cbranch"):
                self.kind, self.target = "cjump", ops.strip()
            elif mnem == "s_endpgm":
                self.kind = "end"
            elif mnem in ("s_setpc_b64", "s_swappc_b64", "s_call_b64", "s_cbranch_g_fork", "s_cbranch_join"):
                raise RuntimeError("hazard pass: indirect control flow (%s) is not handled" % mnem)
            elif mnem not in NO_SDST and not mnem.startswith(("s_cmp", "s_bitcmp", "s_waitcnt")) and parts:
                self.salu_writes = sregs(parts[0])


def patch_function(lines, window):
    """lines: the text lines of one kernel (label line excluded).  Returns (new lines, copies inserted, s_nop expanded)."""
    # ---- items and basic blocks
    items = []                                   # (kind, payload): ("label", name) | ("ins", Ins) | ("other", text)
    for ln in lines:
        body = ln.split(";")[0].rstrip()
        st = body.strip()
        if not st:
            items.append(("other", ln)); continue
        m = re.match(r"^([.\w$]+):", st)
        if m and not body.startswith("\t"):
            items.append(("label", m.group(1), ln)); continue
        if st.startswith("."):
            items.append(("other", ln)); continue
        mnem, _, ops = st.partition(" ")
        if not re.match(r"^[a-z]", mnem):
            items.append(("other", ln)); continue
        items.append(("ins", Ins(ln, mnem, ops.strip())))
    blocks, cur, label_block = [], [], {}
    def close():
        nonlocal cur
        blocks.append(cur); cur = []
    for it in items:
        if it[0] == "label":
            if cur:
                close()
            label_block[it[1]] = len(blocks)      # the block that starts here (possibly empty of instructions)
            cur.append(it)
        else:
            cur.append(it)
            if it[0] == "ins" and it[1].kind in ("jump", "cjump", "end"):
                close()
    if cur:
        close()
    nb = len(blocks)
    succ = [[] for _ in range(nb)]
    for b, blk in enumerate(blocks):
        last = next((it[1] for it in reversed(blk) if it[0] == "ins"), None)
        if last is not None and last.kind in ("jump", "cjump"):
            if last.target not in label_block:
                raise RuntimeError("hazard pass: branch to unknown label " + last.target)
            succ[b].append(label_block[last.target])
        if (last is None or last.kind in ("plain", "cjump")) and b + 1 < nb:
            succ[b].append(b + 1)

    expand_nops = any(it[0] == "ins" and it[1].mnem == "s_wakeup" for it in items)
    n_expanded = [0]

    def transfer(state, blk, emit=None):
        age = list(state)
        n = 0
        for it in blk:
            if it[0] != "ins":
                if emit is not None:
                    emit.append(it[-1])
                continue
            ins = it[1]
            hot = sorted(r for r in ins.reads_addr if age[r] < window)
            if hot and emit is not None:
                for r in hot:
                    emit.append("\ts_mov_b32 %s, %s    ; hazard pass: wait for the VALU write of this register\n" % (reg_name(r), reg_name(r)))
                n += len(hot)
            if hot:                                # (the copies: one wait state each, and the registers are SALU-written now)
                for r in hot:
                    age[r] = window
                age = [min(window, a + len(hot)) for a in age]
            if emit is not None:
                if expand_nops and ins.mnem == "s_nop" and ins.ws > 1:
                    # s_wakeup from another wave of the workgroup ENDS an s_nop in progress, like an s_sleep: the wait
                    # states a single `s_nop N` stands for are not guaranteed in a kernel whose waves wake each other.
                    # N + 1 separate instructions are: each takes its own issue slot.
                    emit.extend("\ts_nop 0" + ("    ; hazard pass: was `%s`" % ins.line.split(";")[0].strip() if k == 0 else "") + "\n"
                                for k in range(ins.ws))
                    n_expanded[0] += 1
                else:
                    emit.append(ins.line)
            age = [min(window, a + ins.ws) for a in age]
            for r in ins.salu_writes:
                age[r] = window
            for r in ins.valu_writes:
                age[r] = 0
        return age, n

    # ---- fixed point of the block-entry states (minimum over predecessors); the kernel starts with every register safe.
    # Computed WITHOUT the copies: they only ever make registers older, so the states stay conservative.
    def transfer_plain(state, blk):
        age = list(state)
        for it in blk:
            if it[0] != "ins":
                continue
            ins = it[1]
            age = [min(window, a + ins.ws) for a in age]
            for r in ins.salu_writes:
                age[r] = window
            for r in ins.valu_writes:
                age[r] = 0
        return age
    entry = [None] * nb
    entry[0] = [window] * NREG
    work = [0]
    while work:
        b = work.pop()
        out = transfer_plain(entry[b], blocks[b])
        for s in succ[b]:
            if entry[s] is None:
                entry[s] = list(out); work.append(s)
            else:
                merged = [min(x, y) for x, y in zip(entry[s], out)]
                if merged != entry[s]:
                    entry[s] = merged; work.append(s)
    new, total = [], 0
    for b, blk in enumerate(blocks):
        st = entry[b] if entry[b] is not None else [window] * NREG      # unreachable code: leave as it is
        _, n = transfer(st, blk, new)
        total += n
    return new, total, n_expanded[0]


def patch_asm(text, window=WINDOW, nops=None):
    """Insert the copies into every kernel of a device assembly file (and split the multi-state s_nop of kernels that use
    s_wakeup).  Returns (text, {kernel: copies}); `nops`, a dict, receives {kernel: s_nop instructions split}."""
    lines = text.splitlines(keepends=True)
    out, report = [], {}
    i = 0
    while i < len(lines):
        m = re.match(r"^([A-Za-z_][\w$]*):", lines[i])
        if m and i > 0 and any(".type\t%s,@function" % m.group(1) in l for l in lines[max(0, i - 8):i]):
            name = m.group(1)
            j = i + 1
            while j < len(lines) and not re.match(r"^\.Lfunc_end\d+:", lines[j]):
                j += 1
            body, n, nx = patch_function(lines[i + 1:j], window)
            out.append(lines[i]); out.extend(body)
            report[name] = n
            if nops is not None:
                nops[name] = nx
            i = j
        else:
            out.append(lines[i]); i += 1
    return "".join(out), report


def run(cmd, verbose, **kw):
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, **kw)


def build_library(out_path, source, flags=(), window=WINDOW, verbose=False, keep_asm=None):
    """source (.hip) -> out_path (.so) with the pass applied to the device code.  Returns {kernel: copies}."""
    src, out_path = os.path.abspath(source), os.path.abspath(out_path)
    with tempfile.TemporaryDirectory() as d:
        asm, pasm = os.path.join(d, "dev.s"), os.path.join(d, "dev_patched.s")
        obj, co, fb = os.path.join(d, "dev.o"), os.path.join(d, "dev.co"), os.path.join(d, "dev.hipfb")
        common = ["-O3", "--offload-arch=" + ARCH, "-std=c++17", "-Wno-unused-command-line-argument", GUARD] + list(flags)
        run(["hipcc"] + common + ["-S", "--cuda-device-only", "-o", asm, src], verbose, cwd=os.path.dirname(src))
        nops = {}
        with open(asm) as f:
            text, report = patch_asm(f.read(), window, nops)
        if verbose:
            print("hazard pass: s_nop split:", {k: v for k, v in nops.items() if v} or "none")
        with open(pasm, "w") as f:
            f.write(text)
        if keep_asm:
            with open(os.path.abspath(keep_asm), "w") as f:
                f.write(text)
        run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + ARCH, "-c", pasm, "-o", obj], verbose)
        run([os.path.join(LLVM, "ld.lld"), "-shared", obj, "-o", co], verbose)
        run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--" + ARCH,
             "-input=/dev/null", "-input=" + co, "-output=" + fb], verbose)
        run(["hipcc"] + common + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
                                  "-fPIC", "-shared", "-o", out_path, src], verbose, cwd=os.path.dirname(src))
    return report


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("source")
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--window", type=int, default=WINDOW)
    ap.add_argument("--keep-asm")
    ap.add_argument("-v", "--verbose", action="store_true")
    a, extra = ap.parse_known_args()
    rep = build_library(a.output, a.source, extra, a.window, a.verbose, a.keep_asm)
    print("hazard pass:", ", ".join("%s +%d" % (k.split("ENS")[0][-24:], v) for k, v in rep.items() if v) or "nothing to insert")


if __name__ == "__main__":
    sys.exit(main())
