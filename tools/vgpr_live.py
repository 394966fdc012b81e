#!/usr/bin/env python3
"""VGPR liveness over a gfx950 assembly listing (hipcc -S --cuda-device-only [-gline-tables-only]): where a kernel's register pressure
peaks. The register allocator only reports totals (-Rpass-analysis=kernel-resource-usage); this walks the function's control-flow graph
backwards and prints, per source line (.loc) or per basic block, how many VGPRs are live there -- spill reloads included, so a kernel
that already spills shows where the allocator ran out.

usage: vgpr_live.py file.s kernel-name-substring [--top N] [--by-block]
Conservative where it matters little: a write under a partial EXEC mask is treated as a full definition (the kernels here branch
wave-uniformly); v_writelane / v_fmac / v_dot*c / SDWA / op_sel destinations also count as uses."""
import re
import sys
from collections import defaultdict

NO_DEF = ("global_store", "buffer_store", "ds_write", "ds_store", "scratch_store", "flat_store", "v_cmp", "v_cmpx", "global_load_lds", "buffer_load_lds",
          "s_", "v_readlane", "v_readfirstlane", "ds_gws", "buffer_wbl2", "buffer_inv", "v_nop", "ds_nop", "global_atomic", "buffer_atomic", "ds_add_u32", "ds_max")
DEF_IS_USE = ("v_writelane", "v_fmac", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c", "v_pk_fmac", "v_fmaak", "v_mfma", "v_smfmac", "v_movrel", "v_cndmask_b16")
BOTH = ("v_swap", "v_permlane16_swap", "v_permlane32_swap")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    by_block = "--by-block" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z][\w$.]*:", l) and name in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
    blocks, cur, loc = [], None, None   # block: {label, ins: [(mn, defs, uses, loc)], succ labels, fall}
    label_of = {}
    for l in lines[start:end]:
        s = l.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\S+|_Z\S+):", s)
        if m or cur is None:
            cur = {"label": m.group(1) if m else "entry", "ins": [], "succ": [], "fall": True}
            label_of[cur["label"]] = len(blocks)
            blocks.append(cur)
            if m:
                continue
        if s.startswith(".loc"):
            p = s.split()
            loc = (files.get(int(p[1]), p[1]), int(p[2]))
            continue
        if s.startswith("."):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        mn, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if mn.startswith("s_cbranch") or mn == "s_branch":
            cur["succ"].append(ops[0])
            if mn == "s_branch":
                cur["fall"] = False
            nb = {"label": f"{cur['label']}+", "ins": [], "succ": [], "fall": True}
            if mn != "s_branch":
                pass
            label_of[nb["label"] + str(len(blocks))] = len(blocks)
            blocks.append(nb)
            prev = cur
            cur = nb
            prev["next"] = len(blocks) - 1
            continue
        if mn in ("s_endpgm", "s_setpc_b64"):
            cur["fall"] = False
        defs, uses = set(), set()
        if any(mn.startswith(p) for p in BOTH):
            defs = regs(ops[0]) | regs(ops[1]); uses = set(defs)
        elif any(mn.startswith(p) for p in NO_DEF):
            for o in ops:
                uses |= regs(o)
        else:
            if ops:
                defs = regs(ops[0])
            for o in ops[1:]:
                uses |= regs(o)
            if any(mn.startswith(p) for p in DEF_IS_USE) or "sdwa" in mn or "op_sel" in rest or mn.endswith("_d16_hi") or "d16" in mn:
                uses |= defs
            if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32") or mn.startswith("v_add_co") or mn.startswith("v_sub_co") or mn.startswith("v_addc_co") or mn.startswith("v_subb_co") or mn.startswith("v_div_scale"):
                pass   # (second destination is an SGPR pair / vcc)
        cur["ins"].append((mn, defs, uses, loc))
    n = len(blocks)
    succ = []
    for i, b in enumerate(blocks):
        s = [label_of[t] for t in b["succ"] if t in label_of]
        if b["fall"] and i + 1 < n:
            s.append(i + 1)
        succ.append(s)
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            live = set()
            for s in succ[i]:
                live |= live_in[s]
            for mn, d, u, _ in reversed(blocks[i]["ins"]):
                live = (live - d) | u
            if live != live_in[i]:
                live_in[i] = live
                changed = True
    per_loc, per_block = defaultdict(int), []
    for i in range(n):
        live = set()
        for s in succ[i]:
            live |= live_in[s]
        mx, mxloc = len(live), None
        for mn, d, u, lc in reversed(blocks[i]["ins"]):
            live = (live - d) | u
            c = len(live | d)
            if c > per_loc[lc]:
                per_loc[lc] = c
            if c > mx:
                mx, mxloc = c, lc
        per_block.append((mx, blocks[i]["label"], mxloc, len(blocks[i]["ins"])))
    print(f"function lines {start}..{end}, {n} blocks, peak live VGPR+AGPR = {max(p[0] for p in per_block)}")
    if by_block:
        for mx, lb, lc, k in sorted(per_block, reverse=True)[:top]:
            print(f"  {mx:4d} live in {lb} ({k} instructions), peak at source line {lc}")
    else:
        for lc, c in sorted(per_loc.items(), key=lambda kv: -kv[1])[:top]:
            print(f"  {c:4d} live at {lc}")


if __name__ == "__main__":
    main()
