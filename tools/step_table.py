"""profiles/rNN_<workload>_table.txt: one replayed step by kernel - time from the replayed-step timeline, HBM bytes and EXECUTED
matrix FLOP (64 x SQ_VALU_MFMA_BUSY_CYCLES) from the PMC passes, reference-formulation FLOPs from bench.kernel_alg_flops.
usage: python tools/step_table.py <round prefix, e.g. r04> <m1|c4>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
R, W = sys.argv[1], sys.argv[2]
tl = open(os.path.join(ROOT, "profiles", "%s_%s_step_timeline.txt" % (R, W))).read().splitlines()
pm = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (R, W))))
by = pm["_step"]["by_kernel"]
ex = {v["kernel"]: v.get("mfma_flop_per_launch") for k, v in pm.items() if not k.startswith("_")}
S, A, To, Tp = bench.WORKLOADS[W]
B, P = S * A, S * A * A
agg = {}
for l in tl[2:]:
    m = re.match(r"^(.*?)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s+(\d+)\s*$", l)
    if m:
        a = agg.setdefault(m.group(1).strip().replace("void ", "").split("<")[0], [0, 0.0])
        a[0] += 1
        a[1] += float(m.group(3))
kfl = bench.kernel_alg_flops(B, P, To, Tp, one_launch_d="disc_update_kernel" in agg)
tot = sum(v[1] for v in agg.values())
out = ["%s (%d scenes x %d agents x %d+%d): one replayed step by kernel - launches, time, HBM bytes (2 FETCH + WRITE, PMC pass), achieved"
       % (W, S, A, To, Tp),
       "HBM rate; `ref GF` / `ref` = algorithmic GFLOP of the REFERENCE formulation (bench.kernel_alg_flops) and its fraction of the",
       "157.3 TFLOP/s fp32 peak - this CREDITS work the kernels eliminate algebraically (fc.4 of the pair embedder is never run per",
       "pair, W_ih . W_embed and fc4 . fc3 are composed) and can exceed 1; `exec GF` / `exec` = matrix FLOP the kernel really issued",
       "(64 x SQ_VALU_MFMA_BUSY_CYCLES, SQ pass) and ITS fraction of the peak = matrix-pipe use, never above 1 (VALU arithmetic is",
       "not in it).  Sources: %s_%s_step_timeline.txt, %s_pmc_%s.json (tools/step_table.py)." % (R, W, R, W), "",
       "%-28s %3s %9s %6s %9s %8s %8s %6s %8s %6s" % ("kernel", "n", "us", "%", "HBM MB", "GB/s", "ref GF", "ref", "exec GF", "exec")]
ex_tot = 0.0
for k, (n, us) in sorted(agg.items(), key=lambda x: -x[1][1]):
    mb, gf, e = by.get(k, 0) / 1e6, kfl.get(k), ex.get(k)
    e_step = e * n if e is not None else None
    ex_tot += e_step or 0.0
    out.append("%-28s %3d %9.1f %5.1f%% %9.1f %8.0f %8s %6s %8s %6s"
               % (k, n, us, 100 * us / tot, mb, mb * 1e6 / (us * 1e-6) / 1e9,
                  "%.2f" % (gf / 1e9) if gf else "-", "%.2f" % (gf / (us * 1e-6) / 157.3e12) if gf else "-",
                  "%.2f" % (e_step / 1e9) if e_step is not None else "-",
                  "%.2f" % (e_step / (us * 1e-6) / 157.3e12) if e_step is not None else "-"))
hb = pm["_step"]["hbm_bytes_per_step"]
fl = bench.alg_flops(B, P, To, Tp)["step"]
out.append("%-28s %3d %9.1f %5.1f%% %9.1f %8.0f %8.2f %6.2f %8.2f %6.2f"
           % ("step", sum(v[0] for v in agg.values()), tot, 100.0, hb / 1e6, hb / (tot * 1e-6) / 1e9, fl / 1e9,
              fl / (tot * 1e-6) / 157.3e12, ex_tot / 1e9, ex_tot / (tot * 1e-6) / 157.3e12))
open(os.path.join(ROOT, "profiles", "%s_%s_table.txt" % (R, W)), "w").write("\n".join(out) + "\n")
print("\n".join(out))
