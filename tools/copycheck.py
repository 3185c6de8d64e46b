#!/usr/bin/env python3
"""Line-overlap check of a source file against the reference tree (container-only development aid: /root/reference does
not exist on the GPU box and nothing in the product or the tests imports this).

    python tools/copycheck.py include/rocalution/solvers.hpp [--by-class] [--show CLASS]

A line counts when, NORMALISED, it is >= 20 characters and not a comment; it is "verbatim" when the same normalised line
occurs anywhere under /root/reference/src.  Normalisation (round 3; the round-2 version compared exact lines and a rename
hid a copy): whitespace and `this->` removed, and on BOTH sides the naming conventions are folded onto one spelling --
member prefixes / suffixes (m_foo, foo_ -> foo), the helpers this repo uses for the reference's idioms (RAMD_EXPECT -> assert,
num<ValueType>( -> static_cast<ValueType>(, say( -> LOG_INFO( ), hook names (doFoo -> Foo, Foo_ -> Foo) and the k-prefixed
local vector names (kr -> r).  Reported per class (text between `class X` headers): share of
verbatim lines and the longest in-order run of consecutive counted lines that are consecutive counted lines of ONE
reference file.
"""
import os
import re
import sys

REF = "/root/reference/src"


def norm(l):
    l = l.replace("this->", "")
    l = re.sub(r"\bRAMD_EXPECT\(", "assert(", l)
    l = re.sub(r"\bnum<(\w+)>\(", r"static_cast<\1>(", l)
    l = re.sub(r"\bm_(\w+)", r"\1", l)          # m_foo -> foo
    l = re.sub(r"\b([A-Za-z]\w*?)_\b", r"\1", l)  # foo_ -> foo
    l = re.sub(r"\bdo([A-Z]\w*)", r"\1", l)      # doSolvePrecond -> SolvePrecond
    l = re.sub(r"\bk([a-z]\w{0,2})\b", r"\1", l)  # kr, kz, kp, kq, kv, kt -> r, z, ...
    return re.sub(r"\s+", "", l)


def counted(l):
    s = l.strip()
    if s.startswith("//") or s.startswith("/*") or s.startswith("*"):
        return None
    s = norm(s)
    return s if len(s) >= 20 else None


def load_ref():
    files = {}
    for r, _, fs in os.walk(REF):
        for f in fs:
            if f.endswith((".cpp", ".hpp", ".h")):
                p = os.path.join(r, f)
                try:
                    lines = open(p, errors="replace").read().splitlines()
                except OSError:
                    continue
                files[p] = [c for c in (counted(l) for l in lines) if c]
    return files


def main():
    path = sys.argv[1]
    show = sys.argv[sys.argv.index("--show") + 1] if "--show" in sys.argv else None
    ref = load_ref()
    allset = set()
    nxt = {}  # (line) -> set of (file, idx)
    for f, ls in ref.items():
        for i, l in enumerate(ls):
            allset.add(l)
            nxt.setdefault(l, []).append((f, i))
    src = open(path).read().splitlines()
    # split by class
    segs, cur, name = [], [], "<preamble>"
    for l in src:
        m = re.match(r"^(class|struct)\s+(\w+)", l)
        if m and not l.rstrip().endswith(";"):
            segs.append((name, cur))
            name, cur = m.group(2), []
        cur.append(l)
    segs.append((name, cur))
    tot_c = tot_v = 0
    print("%-28s %6s %6s %6s %5s" % ("class", "lines", "verb", "share", "run"))
    for name, ls in segs:
        cl = [c for c in (counted(l) for l in ls) if c]
        if not cl:
            continue
        v = [c in allset for c in cl]
        # longest in-order run within one reference file
        best = 0
        for i, c in enumerate(cl):
            for (f, j) in nxt.get(c, ())[:50]:
                k = 0
                rl = ref[f]
                while i + k < len(cl) and j + k < len(rl) and cl[i + k] == rl[j + k]:
                    k += 1
                best = max(best, k)
        tot_c += len(cl)
        tot_v += sum(v)
        print("%-28s %6d %6d %5.0f%% %5d" % (name, len(cl), sum(v), 100.0 * sum(v) / len(cl), best))
        if show == name:
            for l in ls:
                c = counted(l)
                if c and c in allset:
                    print("    | " + l.strip())
    print("%-28s %6d %6d %5.0f%%" % ("TOTAL", tot_c, tot_v, 100.0 * tot_v / max(tot_c, 1)))


if __name__ == "__main__":
    main()
