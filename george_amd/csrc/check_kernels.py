#!/usr/bin/env python3
"""Build-time gate for kernels whose LDS reads are inline assembly with hand-counted waits.

`GH_SP_READ8` / `GH_SP_WAIT` (gh_gemm_tile.h) issue `ds_read_b128` from inline asm and wait for them with a
hand-written `s_waitcnt lgkmcnt(n)` further down.  To the compiler the destination registers are "written" when the
asm statement is issued: if it spills or copies one of them between the read and the wait, the copy holds garbage
(the hardware has no interlock on a VGPR with an LDS return pending).  A kernel that fits its registers is never
treated that way; nothing in the language says so.  This script makes the build say it.  It reads, for every
translation unit, the compiler's resource remarks (`-Rpass-analysis=kernel-resource-usage`) and the device assembly
(`--save-temps`) and fails when

 1. a kernel that contains an inline-asm `ds_read` uses scratch memory or spills a register, or
 2. in ANY kernel, an instruction touches a VGPR while an LDS read into it is still outstanding, i.e. before an
    `s_waitcnt lgkmcnt(n)` has retired the read (LDS operations return in order; scalar-memory loads do not, so while
    one of those is in flight only `lgkmcnt(0)` retires anything).  Every path through the kernel's branches is walked.

Usage:  check_kernels.py <build dir>            (every <tu>.remarks + <tu>-hip-amdgcn-*.s in it)
        check_kernels.py --asm f.s --remarks f.remarks [--must-cover name-substring ...]
Exit status 0 = all kernels pass; 1 = a violation (printed); 2 = inputs missing.  Writes <build dir>/kernel_gate.txt.
"""
import glob
import os
import re
import sys

RE_FUNC = re.compile(r"Function Name: (\S+)")
RE_FIELD = re.compile(r"remark: [^ ]+ +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass")
RE_VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
RE_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
MAX_PENDING = 24


def parse_remarks(path):
    """{kernel: {"scratch": int, "vgpr_spill": int, "sgpr_spill": int, "vgprs": int}}"""
    out, cur = {}, None
    with open(path) as f:
        for line in f:
            if "kernel-resource-usage" not in line:
                continue
            m = RE_FUNC.search(line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            if cur is None:
                continue
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m:
                cur["scratch"] = int(m.group(1))
            m = re.search(r"VGPRs Spill: (\d+)", line)
            if m:
                cur["vgpr_spill"] = int(m.group(1))
            m = re.search(r"SGPRs Spill: (\d+)", line)
            if m:
                cur["sgpr_spill"] = int(m.group(1))
            m = re.search(r" VGPRs: (\d+)", line)
            if m:
                cur["vgprs"] = int(m.group(1))
    return out


def split_functions(path):
    """{name: [lines]} for every function of the device assembly (label `name:` .. `.Lfunc_end`)."""
    funcs, name, body = {}, None, []
    with open(path) as f:
        for raw in f:
            line = raw.rstrip("\n")
            if name is None:
                m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
                if m and not m.group(1).startswith(".L"):
                    name, body = m.group(1), []
                continue
            if line.startswith(".Lfunc_end"):
                funcs[name] = body
                name = None
                continue
            body.append(line)
    return funcs


def vregs(text):
    s = set()
    for m in RE_VREG.finditer(text):
        if m.group(3) is not None:
            s.add(int(m.group(3)))
        else:
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return s


def parse_body(lines):
    """-> (instrs, labels): instrs = [(mnemonic, operand text, in_inline_asm)], labels = {name: index}"""
    instrs, labels, in_app = [], {}, False
    for line in lines:
        code = line.split(";", 1)[0].strip() if not line.lstrip().startswith(";") else ""
        t = line.strip()
        if t.startswith((";;#ASMSTART", ";APP")):
            in_app = True
            continue
        if t.startswith((";;#ASMEND", ";NO_APP")):
            in_app = False
            continue
        if not code:
            continue
        m = re.match(r"^([.\w$]+):$", code)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if code.startswith("."):
            continue
        parts = code.split(None, 1)
        instrs.append((parts[0], parts[1] if len(parts) > 1 else "", in_app))
    return instrs, labels


def lgkm_kind(mn):
    """None = not counted by lgkmcnt; 'lds' = in-order LDS op; 'smem' = out-of-order scalar memory / message"""
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_sendmsg", "s_dcache", "s_scratch_load",
                      "s_atc_probe", "s_store_", "s_buffer_store_", "s_atomic_", "s_buffer_atomic_")):
        return "smem"
    if mn.startswith("flat_") and not mn.startswith(("flat_store",)):
        return "lds"          # flat loads count in both counters; their VGPR destinations are guarded by vmcnt as well
    return None


def lds_dest(mn, ops):
    """VGPRs an LDS instruction writes when its data returns (first operand of every returning DS op)."""
    if not mn.startswith("ds_"):
        return frozenset()
    if mn.startswith(("ds_write", "ds_store", "ds_nop", "ds_gws")):
        return frozenset()
    returning = ("read", "load", "permute", "swizzle", "rtn", "pop", "consume", "append", "ordered_count", "condxchg", "wrap")
    if not any(k in mn for k in returning):
        return frozenset()
    first = ops.split(",", 1)[0]
    return frozenset(vregs(first))


def lint(name, lines):
    """Walk every path; return a list of violation strings."""
    instrs, labels = parse_body(lines)
    n = len(instrs)
    seen, work, bad = set(), [(0, ())], []
    reported = set()
    while work:
        pc, pend = work.pop()
        while pc < n:
            key = (pc, pend)
            if key in seen:
                break
            seen.add(key)
            mn, ops, in_app = instrs[pc]
            if mn == "s_endpgm":
                break
            if mn == "s_waitcnt":
                m = RE_LGKM.search(ops)
                if m:
                    k = int(m.group(1))
                    if k == 0:
                        pend = ()
                    elif not any(e[0] for e in pend):
                        pend = pend[len(pend) - k:] if k < len(pend) else pend
                elif re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", ops):
                    v = int(ops.strip(), 0)                      # raw immediate: lgkmcnt = bits 11:8
                    if ((v >> 8) & 0xF) == 0:
                        pend = ()
                pc += 1
                continue
            touched = vregs(ops)
            if pend and touched:
                busy = set()
                for e in pend:
                    busy |= e[1]
                hit = touched & busy
                if hit and pc not in reported:
                    reported.add(pc)
                    bad.append("%s: instruction %d `%s %s` touches v%s while an LDS read into it is outstanding"
                               % (name, pc, mn, ops, sorted(hit)))
            kind = lgkm_kind(mn)
            if kind == "lds":
                pend = pend + ((False, lds_dest(mn, ops)),)
            elif kind == "smem":
                pend = pend + ((True, frozenset()),)
            # (bounded state: the oldest entries that hold no register and are in-order can go -- retirement is oldest-first,
            #  so which reads `lgkmcnt(k)` leaves pending does not change; beyond that the two oldest entries are merged,
            #  which only makes a register stay "busy" longer)
            while pend and not pend[0][0] and not pend[0][1]:
                pend = pend[1:]
            while len(pend) > MAX_PENDING:
                a, b = pend[0], pend[1]
                pend = ((a[0] or b[0], a[1] | b[1]),) + pend[2:]
            if mn == "s_branch":
                tgt = ops.strip()
                if tgt not in labels:
                    bad.append("%s: branch to unknown label %s" % (name, tgt))
                    break
                pc = labels[tgt]
                continue
            if mn.startswith("s_cbranch"):
                tgt = ops.split(",")[-1].strip()
                if tgt in labels:
                    work.append((labels[tgt], pend))
                else:
                    bad.append("%s: branch to unknown label %s" % (name, tgt))
            if mn in ("s_setpc_b64", "s_swappc_b64"):
                break                                           # (calls: none in this library; the walk ends here)
            pc += 1
    return bad


def has_inline_lds_read(lines):
    in_app = False
    for line in lines:
        t = line.strip()
        if t.startswith((";;#ASMSTART", ";APP")):
            in_app = True
        elif t.startswith((";;#ASMEND", ";NO_APP")):
            in_app = False
        elif in_app and t.startswith("ds_read"):
            return True
    return False


def check(asm_path, remarks_path, report):
    res = parse_remarks(remarks_path)
    funcs = split_functions(asm_path)
    bad, covered = [], []
    for name, lines in funcs.items():
        inline = has_inline_lds_read(lines)
        r = res.get(name)
        if inline:
            covered.append(name)
            if r is None:
                bad.append("%s: contains inline-asm LDS reads but the compiler reported no resource usage for it" % name)
            elif r.get("scratch", 1) != 0 or r.get("vgpr_spill", 1) != 0:
                bad.append("%s: contains inline-asm LDS reads and uses scratch memory (%s bytes/lane, %s VGPRs spilled): "
                           "the compiler may move a read's destination before the data has arrived"
                           % (name, r.get("scratch"), r.get("vgpr_spill")))
        if r is not None or inline:                      # kernels (device functions are inlined; leftovers have no remark)
            bad += lint(name, lines)
            report.append("%-100s %s vgprs=%s scratch=%s%s" % (name[:100], "ok " if not any(b.startswith(name + ":") for b in bad) else "BAD",
                                                               (r or {}).get("vgprs"), (r or {}).get("scratch"),
                                                               "  [inline LDS reads]" if inline else ""))
    return bad, covered


def main(argv):
    must = []
    pairs = []
    out_dir = None
    if len(argv) >= 2 and not argv[1].startswith("--"):
        out_dir = argv[1]
        for rem in sorted(glob.glob(os.path.join(out_dir, "*.remarks"))):
            tu = os.path.basename(rem)[:-len(".remarks")]
            asm = glob.glob(os.path.join(out_dir, tu + "-hip-amdgcn-*.s"))
            if not asm:
                print("check_kernels: no device assembly for %s in %s" % (tu, out_dir))
                return 2
            pairs.append((asm[0], rem))
        must = ["gemm_f64_mfma_dma_sp", "hodlr_bmm_nt_kernel"]
    i = 1 if out_dir is None else 2
    asm = rem = None
    while i < len(argv):
        if argv[i] == "--asm":
            asm = argv[i + 1]; i += 2
        elif argv[i] == "--remarks":
            rem = argv[i + 1]; i += 2
        elif argv[i] == "--must-cover":
            must.append(argv[i + 1]); i += 2
        else:
            print(__doc__)
            return 2
    if asm and rem:
        pairs.append((asm, rem))
    if not pairs:
        print("check_kernels: nothing to check")
        return 2
    bad, covered, report = [], [], []
    for a, r in pairs:
        b, c = check(a, r, report)
        bad += b
        covered += c
    for m in must:
        if not any(m in c for c in covered):
            bad.append("no kernel matching `%s` with inline-asm LDS reads was found: the gate does not cover what it was written for" % m)
    text = ["# kernels with inline-asm LDS reads (scratch must be 0): %d" % len(covered)] + ["#   " + c for c in covered] + report
    if out_dir:
        with open(os.path.join(out_dir, "kernel_gate.txt"), "w") as f:
            f.write("\n".join(text + bad) + "\n")
    if bad:
        print("check_kernels: FAILED")
        for b in bad:
            print("  " + b)
        return 1
    print("check_kernels: %d kernels checked, %d with inline-asm LDS reads, all pass" % (len(report), len(covered)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
