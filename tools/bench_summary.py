"""One-line summary of a bench.py run (file argument or stdin: the compact final JSON line, or the `#detail` line / --detail-file object):
rate, step, scan kernel time and fractions, stage split."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
lines = [l for l in txt.splitlines() if l.startswith("{")]
if not lines:
    print("no JSON line"); sys.exit(1)
l = json.loads(lines[-1])
r = l["roofline"]
st = l.get("stage_ms_per_step") or {k + "_ms": v for k, v in (l.get("stage_ms") or {}).items()}
t = l.get("timing") or {}
mm = t.get("min_max_ms") or ([t["ms_per_step"]["min"], t["ms_per_step"]["max"]] if "ms_per_step" in t else [None, None])
print("%s q/s %.0f step %.4f ms [%s .. %s, %s reps%s] | %s %.4f ms accounting %.3f physical %s pipeline %s | valu %s lds %s conflicts %s | stages %s | parity %s recall %s" % (
    l["config"]["name"], l["value"], l["ms_per_step"], mm[0], mm[1], t.get("repetitions"), ", SUSPECT" if t.get("suspect") else "",
    r["kernel"], r["avg_launch_ms"], r.get("accounting_frac", r["frac"]), r.get("physical_frac"), r.get("pipeline_accounting_frac"),
    r.get("valu_busy_frac"), r.get("lds_busy_frac"), r.get("lds_conflict_ratio"),
    {k[:-3]: round(v, 3) for k, v in st.items()},
    None if l.get("parity") is None else (l["parity"].get("ids_bit_exact"), l["parity"].get("encode_codes_bit_exact")), l.get("recall_at_10")))
