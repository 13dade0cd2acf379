"""Compressed schedule of a kernel's instructions from a `-S` listing: loads, waits, MFMAs, LDS ops, barriers, branches.
usage: isa_loop_schedule.py build/x.s <substring of the mangled name> [max lines]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 200
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l]
for start in starts:
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    print("==", lines[start].split(":")[0], "lines", end - start)
    out = []
    for i in range(start, end):
        t = lines[i].strip()
        if re.match(r"(v_mfma|s_waitcnt|global_load|buffer_load|ds_write|ds_read|s_barrier|s_cbranch|s_branch|\.LBB|scratch_)", t):
            out.append((i, t[:64]))
    prev, cnt, n = None, 0, 0
    for i, t in out:
        k = t.split()[0]
        if k == prev and not k.startswith("s_waitcnt") and not k.startswith(".LBB"):
            cnt += 1
            continue
        if cnt:
            print("        ... x%d more" % cnt)
        print(i - start, t)
        prev, cnt = k, 0
        n += 1
        if n > limit:
            print("...")
            break
