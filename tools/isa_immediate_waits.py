"""Per kernel of a `-S` listing: global loads, and how many of them are followed by `s_waitcnt vmcnt(0)` within a few instructions
(a load whose latency is exposed: usually `cond ? load : 0` compiled into a branch per load).  usage: isa_immediate_waits.py build/x.s ..."""
import re, sys
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    rows = []
    for start in starts:
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        ins = [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        loads = [i for i, t in enumerate(ins) if t.startswith(("global_load", "buffer_load"))]
        imm = 0
        for i in loads:
            for t in ins[i + 1:i + 4]:
                if t.startswith(("global_load", "buffer_load")):
                    break
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                    imm += 1
                    break
        if imm:
            rows.append((imm, len(loads), lines[start].split(":")[0]))
    for imm, n, name in sorted(rows, reverse=True)[:40]:
        print("%4d of %4d loads wait at once  %s" % (imm, n, name[:110]))
