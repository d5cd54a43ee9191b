#!/usr/bin/env python3
"""One-off source transform (round 6): take the lab timing hooks out of the PRODUCT kernels.

  * resolves `#ifdef / #ifndef / #else / #endif` on LAB_STAMP, LAB_TL, LAB_TLB and SPLIT_SPREAD as "not defined";
  * deletes the no-op hook macros that remain (`X3L_*`, `LAB_DECL / LAB_MARK / LAB_ADD / LAB_OUT`, `LAB_TL_STAMP`, `TLB_*`) and every
    statement that only invokes one of them.

The hooks compiled to nothing in product builds; the device ISA before and after the transform is compared with
`scripts/lab/strip_lab_hooks.py --check` (hipcc --cuda-device-only -S on both trees).  The phase-timer builds the lab scripts
used (`attn_x3_phases.py`, `attn_bwd_phases.py`, `gemm_timeline.py`, ...) now build from the tree at the last commit that carried
the hooks: `git archive <that commit> acr_wsss_amd/csrc | tar -x -C scripts/lab/_build/hooks` (build_variant.sh -H).

usage: strip_lab_hooks.py file.hip [...]        (rewrites in place)
"""
import re
import sys

MACROS = ("LAB_STAMP", "LAB_TL", "LAB_TLB", "SPLIT_SPREAD")
NOOP = r"(?:X3L_\w+|LABB_\w+|LAB_DECL|LAB_MARK|LAB_ADD|LAB_OUT|LAB_TL_STAMP|LAB_T|TLB_\w+)"


def resolve(lines):
    out, stack = [], []          # stack entries: (ours, keeping) -- ours: a conditional on one of MACROS
    for ln in lines:
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef)\s+(\w+)", s)
        if m:
            if m.group(2) in MACROS:
                stack.append([True, m.group(1) == "ifndef"])
                continue
            stack.append([False, True])
        elif re.match(r"#\s*if\b", s):
            stack.append([False, True])
        elif re.match(r"#\s*else\b", s) and stack and stack[-1][0]:
            stack[-1][1] = not stack[-1][1]
            continue
        elif re.match(r"#\s*endif\b", s):
            top = stack.pop()
            if top[0]:
                continue
        if all(k for o, k in stack if o):
            out.append(ln)
    assert not stack
    return out


def drop_noops(lines):
    out = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if re.match(r"\s*#\s*define\s+" + NOOP + r"\b", ln):
            while ln.rstrip().endswith("\\"):                # continuation lines of the definition
                i += 1
                ln = lines[i]
            i += 1
            continue
        new = re.sub(r"\b" + NOOP + r"\((?:[^()]|\([^()]*\))*\)\s*;[ \t]*", "", ln)
        new = re.sub(r"\bX3L_DECL\s*;[ \t]*|\bLAB_DECL\s*;[ \t]*|\bTLB_DECL\s*;[ \t]*|\bLABB_DECL\s*;[ \t]*", "", new)
        if new != ln and new.strip() == "":
            i += 1
            continue
        out.append(new.rstrip() + "\n" if new != ln else ln)
        i += 1
    return out


def main():
    for path in sys.argv[1:]:
        src = open(path).read().splitlines(keepends=True)
        dst = drop_noops(resolve(src))
        left = [l for l in dst if re.search(r"\b(?:LAB_STAMP|LAB_TL\b|LAB_TLB|SPLIT_SPREAD|X3L_|LAB_DECL|LAB_MARK|LAB_ADD|LAB_OUT|TLB_)", l)]
        open(path, "w").write("".join(dst))
        print("%s: %d -> %d lines; %d lines still mention a hook" % (path, len(src), len(dst), len(left)))
        for l in left:
            print("    " + l.rstrip())


if __name__ == "__main__":
    main()
