"""One-line summary of a bench.py JSON line (file argument or stdin): rate, step, scan kernel time and fractions, stage split."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
lines = [l for l in txt.splitlines() if l.startswith("{")]
if not lines:
    print("no JSON line"); sys.exit(1)
l = json.loads(lines[-1])
r = l["roofline"]
b = r.get("binding", {})
b = b.get("model", b)
print("%s q/s %.0f step %.3f ms | %s %.3f ms accounting %.3f (moved %.2f lds %.2f valu %.2f) | stages %s | parity %s recall %s" % (
    l["config"]["name"], l["value"], l["ms_per_step"], r["kernel"], r["avg_launch_ms"], r.get("accounting_frac", r["frac"]),
    (b.get("hbm_moved_bytes") or {}).get("frac") or 0, (b.get("lds_gather") or {}).get("frac") or 0, (b.get("valu_issue") or {}).get("frac") or 0,
    {k[:-3]: round(v, 3) for k, v in l["stage_ms_per_step"].items()},
    None if l.get("parity") is None else (l["parity"]["ids_bit_exact"], l["parity"].get("encode_codes_bit_exact")), l["recall_at_10"]))
