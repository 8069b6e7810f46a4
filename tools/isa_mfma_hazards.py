"""ISA check for the kernels that issue matrix instructions from inline asm (attn_cross_rows.hip): the compiler's hazard recogniser does not
see an asm statement, so the wait states gfx940+ wants in software around MFMAs are kept by hand in the source — this script reads the
compiled ISA back and reports, per kernel, (1) the closest non-memory writer of an MFMA's A / B / C operand in front of it, (2) the closest
reader of an MFMA's result behind it (within a window, straight-line only).  usage: python tools/isa_mfma_hazards.py <file.s> [name filter]"""
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.match(r"([av])\[(\d+):(\d+)\]$", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"([av])(\d+)$", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def scan(lines, window=24):
    insts = [l.strip() for l in lines if l.strip() and not l.strip().startswith((";", ".", "/"))]
    insts = [l.split(";")[0].strip() for l in insts]
    min_w, min_r, n = 10 ** 9, 10 ** 9, 0
    worst_w = worst_r = None
    for i, l in enumerate(insts):
        if not l.startswith("v_mfma"):
            continue
        n += 1
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        dst = regs(ops[0])
        srcs = [regs(t) for t in ops[1:4]]
        for back in range(1, window):
            if i - back < 0 or insts[i - back].startswith(("s_cbranch", "s_branch", "s_barrier")) or insts[i - back].endswith(":"):
                break
            p = insts[i - back]
            if p.startswith(("v_mfma", "ds_read", "global_load", "buffer_load", "scratch_load", "s_")):
                continue
            m = re.match(r"(\S+)\s+([^,]+)", p)
            if not m:
                continue
            f, d = regs(m.group(2))
            if d and any(f == sf and d & sr for sf, sr in srcs):
                if back < min_w:
                    min_w, worst_w = back, (p, l)
                break
        for fwd in range(1, window):
            if i + fwd >= len(insts) or insts[i + fwd].startswith(("s_cbranch", "s_branch")) or insts[i + fwd].endswith(":"):
                break
            q = insts[i + fwd]
            if q.startswith(("v_mfma", "s_")):
                continue
            toks = re.findall(r"[av]\[\d+:\d+\]|[av]\d+", q.split(None, 1)[1] if " " in q else "")
            if any(regs(t)[0] == dst[0] and regs(t)[1] & dst[1] for t in toks):
                if fwd < min_r:
                    min_r, worst_r = fwd, (l, q)
                break
    return n, min_w, worst_w, min_r, worst_r


def main():
    src = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", src, re.M):
        name = m.group(1)
        if flt not in name:
            continue
        end = src.find(".Lfunc_end", m.end())
        n, mw, ww, mr, wr = scan(src[m.end():end].split("\n"))
        if n:
            out[name] = (n, mw, mr)
            print(name[:70], "mfmas", n, "| closest operand writer", mw if mw < 10 ** 9 else None, ww, "| closest result reader", mr if mr < 10 ** 9 else None, wr)
    return out


if __name__ == "__main__":
    main()
