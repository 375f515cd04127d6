"""Compressed view of one kernel's instruction stream in a hipcc -S listing (run-length encoded opcode sequence).
usage: python tools/isa_view.py file.s <substring of the kernel symbol> [--full]"""
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    full = "--full" in sys.argv
    s = open(path).read().split("\n")
    start = next(i for i, l in enumerate(s) if key in l and not l.startswith("\t") and l.split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
    out = []
    for l in s[start:end]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.startswith(".LBB"):
                out.append("\n" + t)
            continue
        out.append(t if full else t.split()[0])
    if full:
        print("\n".join(out))
        return
    res, prev, c = [], None, 0
    for o in out:
        if o == prev:
            c += 1
        else:
            if prev:
                res.append(f"{prev}x{c}" if c > 1 else prev)
            prev, c = o, 1
    res.append(f"{prev}x{c}" if c > 1 else prev)
    print(" ".join(res))


if __name__ == "__main__":
    main()
