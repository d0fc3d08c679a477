#!/usr/bin/env python3
"""Where do the 8-byte VALU instructions of a kernel's hot loops start?  (on MI355X a stream that mixes full-rate
and half-rate VALU instructions issues at the sum of their rates only while its 8-byte instructions start at 4 mod 8 --
tools/data_ubench.hip.)  Reads a gfx950 object with llvm-objdump and reports, per basic block above a size, the instruction
mix and the phase.
    python tools/code_phase.py build/obj/reads_kernels.o scan_reads_kernelILi5ELi2 [--min 100]"""
import collections, re, subprocess, sys

HALF = ("v_alignbit", "v_addc_co", "v_add_co", "v_subb", "v_sub_co", "v_bfe", "v_lshl_or", "v_lshl_add", "v_lshlrev_b32", "v_lshlrev_b64",
        "v_and_or", "v_or3", "v_add3", "v_bcnt", "v_cmp", "v_min", "v_max", "v_bfi", "v_mov_b32_dpp", "v_cndmask")


def blocks(obj, kern):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:" % re.escape(kern), out, re.M)
    if not m:
        raise SystemExit("no kernel matching %r in %s" % (kern, obj))
    nxt = re.search(r"^[0-9a-f]+ <[^>]+>:", out[m.end():], re.M)
    body = out[m.end(): m.end() + (nxt.start() if nxt else len(out))]
    ins = []
    for l in body.split("\n"):
        mm = re.match(r"\s+(\S.*?)\s+//\s*([0-9A-Fa-f]+):", l)
        if mm:
            ins.append((int(mm.group(2), 16), mm.group(1).strip()))
    sizes = [ins[i + 1][0] - ins[i][0] for i in range(len(ins) - 1)] + [4]
    # branch targets split blocks too
    targets = set()
    for a, t in ins:
        mm = re.search(r"<[^>]*\+0x([0-9a-f]+)>", t)
        if t.startswith(("s_cbranch", "s_branch")) and mm:
            targets.add(int(mm.group(1), 16))
    base = ins[0][0]
    cur = []
    for (a, t), sz in zip(ins, sizes):
        if (a - base) in targets and cur:
            yield cur; cur = []
        cur.append((a, t, sz))
        if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            yield cur; cur = []
    if cur:
        yield cur


def main():
    obj, kern = sys.argv[1], sys.argv[2]
    minlen = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 100
    print(m_name := kern)
    for b in blocks(obj, kern):
        if len(b) < minlen:
            continue
        valu = [x for x in b if x[1].startswith("v_")]
        v8 = [x for x in valu if x[2] == 8]
        good = sum(1 for x in v8 if x[0] % 8 == 4)
        half = sum(1 for x in valu if x[1].startswith(HALF))
        four = collections.Counter(x[1].split()[0] for x in b if x[2] == 4)
        print("block @%x: %d instrs, %d VALU (%d half-rate class), %d of %d 8-byte VALU at 4 mod 8, 4-byte: %s" %
              (b[0][0], len(b), len(valu), half, good, len(v8), dict(four.most_common(8))))


if __name__ == "__main__":
    main()
