"""the per-vector loop restatement (oracle.compute_codes_loop, the "reference-equivalent CPU" mode of bench.py's
cpu_baseline) must be within +-20 % of the REAL reference's wall time on C1 -- like for like, i.e. on the machine where the reference
itself was timed: the build container (tests/golden/make_golden.py stored c1.ref_encode_vec_per_s there).  Run in the build container:
    python tools/validate_loop_restatement.py   -> profiles/loop_restatement_validation.json (bench.py attaches it to cpu_baseline)"""
import json, os, sys, time, platform
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from oracle import lopq_oracle as O
import golden_inputs as gi
z1 = np.load(os.path.join(REPO, "tests", "golden", "c1.npz"))
om1 = O.OracleModel.from_npz(z1)
x1 = gi.c1_inputs()[0][:8192]
best = 0.0
for rep in range(3):
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0 and n1 < len(x1):
        O.compute_codes_loop(om1, x1[n1:n1 + 64]); n1 += 64
    best = max(best, n1 / (time.perf_counter() - t0))
ref = float(z1["ref_encode_vec_per_s"])
out = {"where": "build container (%s, %d CPUs visible)" % (platform.processor() or platform.machine(), os.cpu_count()),
       "c1_encode_loop_vps": best, "c1_encode_reference_vps": ref, "loop_over_reference": best / ref,
       "within_20_percent": bool(abs(best / ref - 1.0) <= 0.2),
       "note": "best of three 4-second runs of the loop restatement, single thread, against the reference's own figure stored with the fixture "
               "(same container, taken when tests/golden/c1.npz was made)"}
json.dump(out, open(os.path.join(REPO, "profiles", "loop_restatement_validation.json"), "w"), indent=1)
print(json.dumps(out))
