#!/usr/bin/env python3
"""Writes profiles/<round>/ROUND_SUMMARY.md from all_configs.txt (times), pmc_summary.json (bytes) and bench.py's byte model.
    python profiles/round_summary.py r03"""
import importlib.util, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
D = os.path.join(ROOT, "profiles", rnd)
rows = {}
for line in open(os.path.join(D, "all_configs.txt")):
    m = re.match(r"--config (\S+)(.*?)\s+ms/step=(\S+) value=(\S+) regions=\[.*?\] (\{.*?\}) frac", line)
    if not m:
        continue
    name = m.group(1) + (" --no-obs" if "--no-obs" in m.group(2) else "") + (" --diffuse" if "--diffuse" in m.group(2) else "")
    rows.setdefault(name, (float(m.group(3)), float(m.group(4)), eval(m.group(5))))
pmc = json.load(open(os.path.join(D, "pmc_summary.json")))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
notes = open(os.path.join(D, "ROUND_NOTES.md")).read() if os.path.exists(os.path.join(D, "ROUND_NOTES.md")) else ""
out = ["# Round %s — one-page summary" % rnd[1:].lstrip("0"), "",
       "Sources: `all_configs.txt` (times, HIP events of that run), `pmc_summary.json` (HBM-side bytes, separate `--pmc` passes of the same",
       "command), `bench.py:algorithmic_bytes` (SURVEY §8(d)'s model).  One MI355X box, product path, final tree of the round, `bench.py --config cX`",
       "with the default 400-step episode ageing.  `alg` = algorithmic bytes per launch, `real` = (2·FETCH_SIZE + WRITE_SIZE)·1024 per launch",
       "(fabric side: Infinity Cache hits are counted).", "",
       "| config | ms/step | ant-steps/s | kernel | ms | alg MB | alg TB/s (frac of 8) | real MB | real TB/s |", "|---|---|---|---|---|---|---|---|---|"]
names = dict(k_update_move="update_move", k_perceive="perceive", k_sweep_sep2="sweep", k_move="move", k_update_one="update")
for cfg in ("c2", "c3", "c4", "c5"):
    if cfg not in rows:
        continue
    ms, val, k = rows[cfg]
    c = bench.CONFIGS[cfg]
    K = 7 if c["R"] else 6
    ab = bench.algorithmic_bytes(c["N"], c["W"], c["H"], 2, K, obs_bytes=2 if cfg == "c5" else 4)
    ab["update_move"] = ab["move"] + ab["update"]
    first = True
    for kn, kms in k.items():
        alg = ab[names[kn]] * c["E"] / 1e6
        d = pmc.get(cfg, {}).get(kn, {})
        real = (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / 1e6 if d else None
        out.append("| %s | %s | %s | `%s` | %.4f | %.0f | %.2f (%.2f) | %s | %s |" % (
            cfg if first else "", "%.4f" % ms if first else "", "%.3e" % val if first else "", kn, kms, alg, alg / kms / 1e3,
            alg / kms / 1e3 / 8, "%.0f" % real if real else "—", "%.2f" % (real / kms / 1e3) if real else "—"))
        first = False
for cfg, label in (("c5 --no-obs", "c5 act-only"), ("c3 --diffuse", "c3 + 3×3 diffusion"), ("c1", "c1")):
    if cfg not in rows:
        continue
    ms, val, k = rows[cfg]
    first = True
    for kn, kms in k.items():
        out.append("| %s | %s | %s | `%s` | %.4f | — | — | — | — |" % (label if first else "", "%.4f" % ms if first else "",
                                                                      "%.3e" % val if first else "", kn, kms))
        first = False
out += ["", notes]
open(os.path.join(D, "ROUND_SUMMARY.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:30]))
