"""Static check of the four-deep watch (persist.h watch4): between '; WATCH4_BEGIN vN' and the matching '; WATCH4_END vN' of a build's
assembly nothing may write vN (reads of the watch are still in flight into it) and vN may not be spilled; the END must name the same
register as the BEGIN, and a full wait (s_waitcnt vmcnt(0): the sweep's, or watch_drain's own) must lie between the two -- an END straight
behind the watch releases the register while reads are in flight. usage: python tools/check_watch_regs.py rwkv.cpp_amd/build/persist_v47-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys


def dests(line):
    """VGPRs an instruction writes (first operand of VALU / VMEM-load / DS-read instructions), as a set of ints"""
    m = re.match(r"\s+([a-z_0-9]+)\s+(.*)", line)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    if op.startswith(("s_", "buffer_store", "global_store", "scratch_store", "ds_write", "ds_store", "flat_store", ";")):
        return set()
    first = rest.split(",")[0].strip()
    out = set()
    m1 = re.match(r"v\[(\d+):(\d+)\]", first)
    if m1:
        out |= set(range(int(m1.group(1)), int(m1.group(2)) + 1))
    m2 = re.match(r"v(\d+)$", first)
    if m2:
        out.add(int(m2.group(1)))
    return out


def check(path):
    bad, sites, kernel = [], 0, None
    open_reg, open_line, waited = None, 0, False
    for n, line in enumerate(open(path), 1):
        if re.match(r"^_Z\w+:", line):
            kernel = line.strip().rstrip(":")
        if open_reg is not None and re.search(r"s_waitcnt vmcnt\(0\)", line):
            waited = True
        m = re.search(r"; WATCH4_BEGIN v(\d+)", line)
        if m:
            waited = False
            # (a BEGIN while one is open: the abort path's re-entry of the loop, same register expected)
            if open_reg is not None and open_reg != int(m.group(1)):
                bad.append((kernel, n, "nested BEGIN with another register"))
            open_reg, open_line = int(m.group(1)), n
            sites += 1
            continue
        m = re.search(r"; WATCH4_END v(\d+)", line)
        if m:
            if open_reg is not None and not waited:
                bad.append((kernel, n, "END without a full wait behind the BEGIN at line %d" % open_line))
            if open_reg is not None and int(m.group(1)) != open_reg:
                bad.append((kernel, n, "END names v%s, BEGIN (line %d) v%d: the value was copied while reads were in flight" % (m.group(1), open_line, open_reg)))
            open_reg = None
            continue
        if open_reg is not None:
            if "s_endpgm" in line:
                open_reg = None
                continue
            if open_reg in dests(line) and "WATCH4" not in line:
                # the watch's own loop re-entry (buffer_load_dword vN ... sc1 inside the asm block) is legitimate
                if not re.search(r"buffer_load_dword v%d, v\d+, s\[\d+:\d+\], 0 offen sc1" % open_reg, line):
                    bad.append((kernel, n, "v%d written inside the watch window: %s" % (open_reg, line.strip())))
    return sites, bad


if __name__ == "__main__":
    sites, bad = check(sys.argv[1])
    print("%d watch sites, %d violations" % (sites, len(bad)))
    for b in bad[:40]:
        print(b)
    sys.exit(1 if bad else 0)
