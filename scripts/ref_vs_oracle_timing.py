#!/usr/bin/env python3
"""
How fast is the CPU oracle (what bench.py's `cpu_baseline` legs time on the GPU box, kind = "port") compared with the
REFERENCE's own Python on the same inputs?  SURVEY 8(d) promises that cross-check; the reference never travels, so it can
only be made here.  RUNS ONLY IN THE AUTHORING CONTAINER (needs /root/reference; the reference modules are imported with
the stub packages of tests/golden/make_goldens.py, numba.jit = identity) and writes profiles/r05_ref_vs_oracle.json --
numbers, no reference source.  One process, one thread (BLAS pools pinned to 1) for both sides.

Legs (seeded synthetic inputs of the BASELINE configs' per-pair shapes):
  simple       Simple.oti + Simple.simple_sim (simple_silva.py:45-54, 68-118) on 12 x ~200 pooled frames, SSLEN 10
               vs oracle.simple_pair
  earlyfusion  the similarity() chain of earlyfusion_traile.py:157-198 on ~400 x 400 blocks: get_csm x2, get_csm_blocked_oti,
               csm_to_binary x4, getWCSM x3 timed from the reference; its smith_waterman_constrained (alignment_tools.py:
               26-46) is numba-jitted in a real installation and PLAIN PYTHON here (0.7 s per matrix), so it is timed once
               and reported separately -- the chain figure uses the oracle's C Smith-Waterman for both sides
  serra09_csm  cross_recurrence.get_csm on the 108-dim stacked frames of a 450 x 450 pair (the only part of Serra09's
               per-pair work the reference repo holds; the rest is essentia) vs the oracle's C chain up to the distances
"""
import json
import os
import sys
import time

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import importlib  # noqa: E402

import make_goldens  # noqa: E402  (install_stubs: the stub packages of SURVEY App. B)
import oracle  # noqa: E402
from acoss_amd import synth  # noqa: E402


def best_of(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), out


def main():
    if not os.path.isdir("/root/reference"):
        raise SystemExit("ref_vs_oracle_timing: /root/reference is not here (authoring container only)")
    make_goldens.install_stubs()
    cr = importlib.import_module("acoss.algorithms.utils.cross_recurrence")
    al = importlib.import_module("acoss.algorithms.utils.alignment_tools")
    sf = importlib.import_module("acoss.algorithms.utils.similarity_fusion")
    ss = importlib.import_module("acoss.algorithms.simple_silva")
    rec = {"host": {"cpu_count": os.cpu_count(), "threads_used": 1, "numpy": np.__version__},
           "note": "reference = /root/reference imported with stub packages (numba.jit = identity); oracle = oracle/ of this repo; "
                   "best of N repetitions each, one thread"}
    rng = np.random.default_rng(2025)

    # ---- SiMPle
    simple = ss.Simple.__new__(ss.Simple)                      # (no constructor: it would want a dataset csv)
    simple.SSLEN = 10
    A = rng.random((12, 206))
    B = rng.random((12, 180))
    t_ref, v_ref = best_of(lambda: -simple.simple_sim(A, simple.oti(A, B)[0]), 5)
    t_orc, v_orc = best_of(lambda: oracle.simple_pair(A, B, 10), 5)
    rec["simple"] = {"shape": "12 x 206 vs 12 x 180 pooled frames, SSLEN 10", "reference_s_per_pair": t_ref, "oracle_s_per_pair": t_orc,
                     "oracle_speed_over_reference": t_ref / t_orc, "values_agree_to": float(abs(v_ref - v_orc))}

    # ---- EarlyFusion chain
    nb1, nb2 = 410, 390
    f1 = {"mfccs": rng.standard_normal((nb1, 650)).astype(np.float32), "ssms": rng.random((nb1, 1225)).astype(np.float32),
          "chromas": rng.random((nb1, 480)).astype(np.float32), "chroma_med": rng.random(12)}
    f2 = {"mfccs": rng.standard_normal((nb2, 650)).astype(np.float32), "ssms": rng.random((nb2, 1225)).astype(np.float32),
          "chromas": rng.random((nb2, 480)).astype(np.float32), "chroma_med": rng.random(12)}
    kappa, K = 0.1, 10

    def ref_chain(sw):
        C = {"mfccs": cr.get_csm(f1["mfccs"], f2["mfccs"]), "ssms": cr.get_csm(f1["ssms"], f2["ssms"]),
             "chromas": cr.get_csm_blocked_oti(f1["chromas"], f2["chromas"], f1["chroma_med"], f2["chroma_med"], cr.get_csm_cosine)}
        sc = {s: sw(cr.csm_to_binary(C[s], kappa)) for s in C}
        W = np.zeros_like(C["mfccs"])
        for s in C:
            W += sf.getWCSM(C[s], K, K)
        sc["early"] = sw(cr.csm_to_binary(np.exp(-W), kappa))
        return sc

    sw_c = lambda Bm: oracle.sw_constrained(np.asarray(Bm, dtype=np.uint8))       # noqa: E731
    t_ref, sc_ref = best_of(lambda: ref_chain(sw_c), 3)
    t_orc, (sc_orc, _) = best_of(lambda: oracle.earlyfusion_pair(f1, f2, kappa, K), 3)
    Bm = np.asarray(cr.csm_to_binary(cr.get_csm(f1["mfccs"], f2["mfccs"]), kappa), dtype=np.uint8)
    t_sw_py, v_py = best_of(lambda: al.smith_waterman_constrained(Bm), 1)
    t_sw_c, v_c = best_of(lambda: oracle.sw_constrained(Bm), 3)
    rec["earlyfusion"] = {"shape": "%d x %d blocks (K = 650 / 1225 / 480), kappa 0.1, K 10" % (nb1, nb2),
                          "reference_chain_s_per_pair_with_oracle_sw": t_ref, "oracle_chain_s_per_pair": t_orc,
                          "oracle_speed_over_reference": t_ref / t_orc,
                          "scores_equal": {s: bool(sc_ref[s] == sc_orc[s]) for s in sc_ref},
                          "smith_waterman_reference_unjitted_s": t_sw_py, "smith_waterman_oracle_c_s": t_sw_c,
                          "smith_waterman_values": [float(v_py), float(v_c)],
                          "note": "the reference's smith_waterman_constrained is @jit(nopython=True) in a real installation; un-jitted here "
                                  "it takes %.2f s per matrix (x 4 per pair) and would make the reference look %.0f x slower than it is -- "
                                  "both chains are therefore timed with the oracle's C Smith-Waterman" % (t_sw_py, 4 * t_sw_py / t_orc)}

    # ---- Serra09: the stacked-frame distance matrix
    d = synth.rand_set(2, T=450, seed=77)
    X = d["frames"][:450].astype(np.float32)
    Y = d["frames"][450:900].astype(np.float32)
    m = 9
    Xe = np.concatenate([X[k:len(X) - m + k] for k in range(m)], axis=1)       # T - m tau stacked frames of 108 bins
    Ye = np.concatenate([Y[k:len(Y) - m + k] for k in range(m)], axis=1)
    t_ref, D_ref = best_of(lambda: cr.get_csm(Xe, Ye), 5)
    t_orc, res = best_of(lambda: oracle.serra09_pair(X, Y, oracle.serra09_params(oti=False), want_intermediates=True), 5)
    rec["serra09_csm"] = {"shape": "450 x 450 pooled frames, m = 9: 441 x 441 distances of 108-dim stacks",
                          "reference_get_csm_s": t_ref, "oracle_whole_pair_s": t_orc,
                          "note": "the reference repo holds only the distance matrix of this chain (the thresholds and Qmax are essentia's); the "
                                  "oracle figure is the WHOLE pair (OTI off, distances, both percentile passes, Qmax) in C"}
    rs, re_ = rec["simple"]["oracle_speed_over_reference"], rec["earlyfusion"]["oracle_speed_over_reference"]
    rec["reading"] = ("cpu_baseline legs of bench.py / bench_other.py time the oracle (kind = 'port').  Per pair and thread the oracle runs at "
                      "%.1f x the speed of the reference's own numpy code for SiMPle (a vectorised restatement of the reference's per-row loop) and at "
                      "%.2f x for the EarlyFusion chain (its csm_to_binary sorts whole rows stably where the reference partitions; C Smith-Waterman on "
                      "both sides): a GPU / CPU ratio quoted against the port UNDERSTATES the ratio against the reference by %.1f x for SiMPle and "
                      "OVERSTATES it by %.1f x for EarlyFusion" % (rs, re_, rs, 1.0 / re_))
    out = os.path.join(ROOT, "profiles", "r05_ref_vs_oracle.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
