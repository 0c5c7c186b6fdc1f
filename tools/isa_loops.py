#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from the build's device assembly (veloxseg_amd/lib/obj/<file>.fixed.s): static counts per back-edge region --
how many VALU / SALU / LDS / global / MFMA / transcendental instructions one trip of each loop issues.  argv: <file stem> <substring of the mangled kernel name> ..."""
import os, re, sys
from collections import Counter


def loops_of(text, name):
    a = text.index(name + ":")
    b = text.index(".Lfunc_end", a)
    L = text[a:b].split("\n")
    labels = {}
    for i, l in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(L):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                out.append((labels[t], i))
    return L, out


def mix(lines):
    ins = [x.strip().split()[0] for x in lines if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    c = Counter(ins)
    tr = sum(v for k, v in c.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k))
    return dict(n=len(ins), mfma=sum(v for k, v in c.items() if "mfma" in k), valu=sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k),
                salu=sum(v for k, v in c.items() if k.startswith("s_")), lds=sum(v for k, v in c.items() if k.startswith("ds_")),
                glob=sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))), trans=tr, div=c.get("v_div_scale_f32", 0) // 2,
                waitcnt=c.get("s_waitcnt", 0), top=c.most_common(12))


if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "veloxseg_amd", "lib", "obj")
    text = open(os.path.join(root, sys.argv[1] + ".fixed.s")).read()
    names = re.findall(r"^(_Z\w+|vx_\w+):", text, re.M)
    for sub in sys.argv[2:]:
        for name in names:
            if sub not in name:
                continue
            L, lp = loops_of(text, name)
            m = mix(L)
            print(f"{name[:90]}\n   whole kernel: {m['n']} instr, mfma {m['mfma']}, valu {m['valu']}, salu {m['salu']}, lds {m['lds']}, global {m['glob']}, transcendental {m['trans']}, ieee div {m['div']}")
            for a0, b0 in sorted(lp, key=lambda t: -(t[1] - t[0]))[:int(os.environ.get("VX_NLOOPS", "4"))]:
                m = mix(L[a0:b0])
                print(f"   loop @{a0}..{b0}: {m['n']} instr | mfma {m['mfma']} valu {m['valu']} salu {m['salu']} lds {m['lds']} global {m['glob']} trans {m['trans']} div {m['div']} waitcnt {m['waitcnt']}")
                print("        " + ", ".join(f"{k} {v}" for k, v in m["top"]))
