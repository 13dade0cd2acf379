"""Randomised GPU-vs-oracle parity: random model shapes, data with many duplicate codes and near-ties, random quota / limit
(all three ranking paths: float32-prefilter scan, float64 scan, segmented sort).  Usage: python tests/tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import lopq_oracle as O
from columbiaimagesearch_amd.lopq import LOPQModel, LOPQModelPCA, LOPQSearcherHIP

# FUZZ_M / FUZZ_K: other shapes, e.g. FUZZ_M=2,6,12,32 FUZZ_K=10,100,256 (the generic kernels: float64 scan, global-table distances)
MS = [int(v) for v in os.environ.get("FUZZ_M", "4,8,16").split(",")]
KS = [int(v) for v in os.environ.get("FUZZ_K", "16,64,256").split(",")]
# >= 256: matrix-core coarse prefilter, tiny cells (k_plan_par, k_adc_direct); slow in the oracle (Python heap over V * V cells)
VS = [int(v) for v in os.environ.get("FUZZ_V", "2,4,16,40,40,300,1024").split(",")]

def run(cases, seed0):
  bad = 0
  only = [int(v) for v in os.environ["FUZZ_ONLY"].split(",")] if os.environ.get("FUZZ_ONLY") else None
  for case in range(cases):
      if only is not None and case not in only:
          continue
      rs = np.random.RandomState(seed0 * 1000 + case)
      t_case = time.time()
      M = int(rs.choice(MS))
      K = int(rs.choice(KS))
      V = int(rs.choice(VS))
      w = int(rs.choice([2, 4, 8]))
      D = M * w
      h, nf = D // 2, M // 2
      dt = rs.choice([np.float32, np.float64])
      Cs = [rs.randn(V, h).astype(dt) for _ in range(2)]
      Rs = [np.stack([np.linalg.qr(rs.randn(h, h))[0] for _ in range(V)]) for _ in range(2)]
      mus = [rs.randn(V, h) * 0.05 for _ in range(2)]
      subs = [[rs.randn(K, w) * rs.choice([0.1, 0.6]) for _ in range(nf)] for _ in range(2)]
      use_pca = rs.rand() < 0.35  # PCA front end (renorm on/off), input wider than the LOPQ space
      if use_pca:
          D_in = D + int(rs.choice([0, 5, 40]))
          P = np.linalg.qr(rs.randn(D_in, D_in))[0][:, :D]
          pmu = rs.randn(D_in) * 0.1
          renorm = bool(rs.rand() < 0.5)
          Cs = [c.astype(np.float32) for c in Cs]  # apply_PCA hands float32 to the coarse stage
          m = LOPQModelPCA(renorm=renorm, parameters=(tuple(Cs), tuple(Rs), tuple(mus), tuple(subs), P, pmu))
          om = O.OracleModel(Cs, Rs, mus, subs, pca_P=P, pca_mu=pmu, renorm=renorm)
      else:
          D_in = D
          m = LOPQModel(parameters=(tuple(Cs), tuple(Rs), tuple(mus), tuple(subs)))
          om = O.OracleModel(Cs, Rs, mus, subs)
      n = int(rs.choice([300, 5000, 60000])) if V < 300 else (int(rs.choice([5000, 60000])) if V < 1024 else 60000)
      base = rs.randn(max(n // int(rs.choice([1, 3, 50])), 1), D_in)  # few distinct points -> many duplicate codes
      X = (base[rs.randint(0, len(base), n)] + rs.choice([0.0, 1e-3, 0.3]) * rs.randn(n, D_in)).astype(dt)
      nq = int(rs.choice([24, 96, 300, 600]))  # >= 64: two queries per workgroup in the scan; 600: past every small-batch threshold
      if V >= 300:
          nq = min(nq, 96)  # the oracle's heap walks thousands of cells per query there
      Q = (X[rs.randint(0, n, nq)] + 0.05 * rs.randn(nq, D_in)).astype(dt)
      coarse, fine = m.predict_batch(X)
      oc, of = O.compute_codes(om, X[:3000])  # encode parity: bit-exact codes (numpy summation order, first minimum)
      if not (np.array_equal(coarse[:3000], oc) and np.array_equal(fine[:3000], of)):
          bad += 1
          print("ENCODE MISMATCH case %d: M=%d K=%d V=%d w=%d dtype=%s: %d coarse, %d fine rows differ" % (
              case, M, K, V, w, np.dtype(dt).name, int((coarse[:3000] != oc).any(1).sum()), int((fine[:3000] != of).any(1).sum())))
      s = LOPQSearcherHIP(m)
      s.add_codes_array(coarse, fine, dedup=False)
      oi = O.OracleCSRIndex(om, coarse, fine)
      quota = int(rs.choice([1, 10, 100, 1000, 10000]))
      if V >= 300:
          quota = max(1, min(quota, 1000, n // 8))  # a quota the index cannot fill walks all V * V cells in the oracle's Python heap
      limit = rs.choice([None, 1, 7, 100, 184, 185, 300, 440, 441, 600, 952, 953, 1000, 4000])
      limit = None if limit is None else int(limit)
      if limit is None and quota > 20000:
          limit = 100
      mode = int(rs.choice([0, 2, 3, 4, 5, 5]))  # automatic routing, float32-prefilter kernel, 16-bit fixed-point kernel (streaming / two-pass / sampled)
      if os.environ.get("FUZZ_ALL_MODES"):  # diagnosis: every route on this case
          for md in (0, 1, 2, 3, 4, 5):
              s.set_scan_mode(mode=md)
              rr = s.search_batch(Q, quota=quota, limit=limit)
              nbad = 0
              for qi in range(nq):
                  ids, dists, visited = oi.search(Q[qi], quota=quota, limit=limit)
                  k = len(ids)
                  if not (rr["n_found"][qi] == k and np.array_equal(rr["ids"][qi, :k], ids)):
                      if nbad == 0:
                          d = np.nonzero(rr["ids"][qi, :k] != ids)[0]
                          print("  mode %d query %d: first diff at rank %d of %d (%d differ); got id %d dist %.17g, want id %d dist %.17g; kernel %s" % (
                              md, qi, d[0] if len(d) else -1, k, len(d), rr["ids"][qi, d[0]] if len(d) else -1, rr["dists"][qi, d[0]] if len(d) else 0,
                              ids[d[0]] if len(d) else -1, dists[d[0]] if len(d) else 0, s.last_stats()["scan_kernel"]))
                      nbad += 1
              print("  mode %d: %d of %d queries differ" % (md, nbad, nq), flush=True)
      if os.environ.get("FUZZ_MODE"):
          mode = int(os.environ["FUZZ_MODE"])
      s.set_scan_mode(mode=mode)
      r = s.search_batch(Q, quota=quota, limit=limit)
      ok = True
      for qi in range(nq):
          ids, dists, visited = oi.search(Q[qi], quota=quota, limit=limit)
          k = len(ids)
          if not (r["n_found"][qi] == k and r["visited"][qi] == visited and np.array_equal(r["ids"][qi, :k], ids)
                  and np.allclose(r["dists"][qi, :k], dists, rtol=1e-9, atol=0)):
              ok = False
              print("MISMATCH case %d query %d: M=%d K=%d V=%d w=%d n=%d nq=%d quota=%d limit=%s dtype=%s mode=%d found %d/%d" % (
                  case, qi, M, K, V, w, n, nq, quota, limit, np.dtype(dt).name, mode, r["n_found"][qi], k))
              break
      if ok and limit is not None and limit <= 3072 and rs.rand() < 0.3:
          # the same index cut into three cell shards: packed partial lists merged == the single index
          import torch
          from columbiaimagesearch_amd.lopq.search import merge_packed_dev
          qd = torch.as_tensor(Q).cuda().contiguous()
          world = 3
          parts, cnts = [], []
          for rk in range(world):
              sh = LOPQSearcherHIP(m, shard=(rk, world))
              sh.add_codes_array(coarse, fine, dedup=False)
              pp = sh.search_partial_packed_dev(qd, quota=quota, limit=limit)
              tot = int(pp["total"].item())
              parts.append(pp["packed"][:tot].clone()); cnts.append(pp["cnt"].clone())
              sh.close()
          stride = max(max(int(p_.shape[0]) for p_ in parts), 1)
          buf = torch.zeros((world, stride, 4), dtype=torch.int64, device="cuda")
          for rk in range(world):
              buf[rk, :parts[rk].shape[0]] = parts[rk]
          cnt = torch.stack(cnts).contiguous()
          off = (torch.cumsum(cnt, dim=1, dtype=torch.int64) - cnt).contiguous()
          o2 = merge_packed_dev(buf, off, cnt, nq, limit)
          torch.cuda.synchronize()
          a, b = o2["dists"].cpu().numpy(), r["dists"]
          if not (np.array_equal(o2["ids"].cpu().numpy(), r["ids"]) and np.array_equal(o2["n_found"].cpu().numpy(), r["n_found"])
                  and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])):
              ok = False
              print("SHARDED MISMATCH case %d: M=%d K=%d V=%d n=%d quota=%d limit=%s" % (case, M, K, V, n, quota, limit))
      bad += 0 if ok else 1
      s.close()
      if os.environ.get("FUZZ_VERBOSE"):
          print("case %d ok=%s M=%d K=%d V=%d w=%d n=%d nq=%d quota=%d limit=%s %.1fs" % (case, ok, M, K, V, w, n, nq, quota, limit, time.time() - t_case), flush=True)
  return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad = run(cases, seed0)
    print("fuzz: %d cases, %d mismatching, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
