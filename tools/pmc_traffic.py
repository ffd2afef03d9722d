"""profiles/rNN_pmc_<workload>.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (separate --pmc runs) + the kernel trace.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and, on gfx950, FETCH_SIZE reports
half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) - validated in round 1 on dec_rollout_bwd, whose
reads (saved LSTM rows + a1/a2) are 58 MB by construction vs 2 x 27.4 MB counted.
`_step` = per-step totals: sum over kernels of (mean bytes per launch) x (launches per step; counted against the one
decode-forward launch every step has).
Optional 4th argument: the SQ pass (SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs).  A SIMD's matrix pipe retires 64 fp32
FLOP per busy cycle (v_mfma_f32_16x16x4_f32: 2048 FLOP per 32-cycle issue slot), so EXECUTED matrix FLOP per launch = 64 x
busy cycles - what the kernels really issued, next to the reference-formulation FLOP counts of bench.kernel_alg_flops (which
credit work the kernels eliminate algebraically).  VALU arithmetic (gate non-linearities, tail columns) is not in it.
    python tools/pmc_traffic.py fetch.db write.db out.json [sq.db]"""
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict


def mean_counter(db, name):
    con = sqlite3.connect(db)
    per, cnt = defaultdict(float), defaultdict(set)
    for k, c, v, d in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if c == name:
            key = k.split("(")[0].replace("void ", "").split("<")[0]
            per[key] += v
            cnt[key].add(d)
    return {k: per[k] / len(cnt[k]) for k in per}, {k: len(cnt[k]) for k in per}


fetch, nf = mean_counter(sys.argv[1], "FETCH_SIZE")
write, nw = mean_counter(sys.argv[2], "WRITE_SIZE")
abi = {"dec_rollout_bwd_kernel": "sw_dec_rollout_bwd", "dec_rollout_fwd_kernel": "sw_dec_rollout_fwd",
       "dec_rollout_fwd2_kernel": "sw_dec_rollout_fwd",
       "enc_lstm_fwd_kernel": "sw_enc_lstm_fwd", "enc_lstm_bwd_kernel": "sw_enc_lstm_bwd",
       "disc_fwd_kernel": "sw_disc_fwd", "disc_bwd_kernel": "sw_disc_bwd", "disc_update_kernel": "sw_disc_update",
       "wgrad_partial_kernel": "wgrad_partial", "wgrad_reduce_kernel": "wgrad_reduce",
       "social_pool_fwd_kernel": "sw_social_pool_fwd", "social_pool_bwd_kernel": "sw_social_pool_bwd",
       "social_pool_bwd_rows_kernel": "sw_social_pool_bwd_rows", "stage_step_kernel": "sw_stage_step",
       "enc_compose_bwd_kernel": "enc_compose_bwd"}
# (the decode forward is ONE launch per step under either of its kernels: 16-agent tiles, or two blocks per workgroup)
steps = max(nf.get("dec_rollout_fwd_kernel", 0) + nf.get("dec_rollout_fwd2_kernel", 0), 1)
mfma, nm = mean_counter(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES") if len(sys.argv) > 4 else ({}, {})
sq_steps = max(nm.get("dec_rollout_fwd_kernel", 0) + nm.get("dec_rollout_fwd2_kernel", 0), 1)
out, step_bytes, step_k, step_mfma = {}, 0.0, {}, 0.0
for k in sorted(set(fetch) | set(write)):
    if "at::" in k or "rocclr" in k or k in ("spin_kernel", "nop_kernel", "traj4d_kernel", "gen_images_kernel", "disc_images_kernel"):
        continue            # torch's own kernels and helpers that are not part of a replayed step
    name = abi.get(k, k)
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    per_launch = (2 * f + w) * 1024
    lps = nf.get(k, nw.get(k, 0)) / steps
    out[name] = {"kernel": k, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                 "hbm_bytes_per_launch": int(per_launch), "launches_per_step": round(lps, 3)}
    step_bytes += per_launch * lps
    step_k[k] = int(per_launch * lps)
    if k in mfma:
        out[name]["mfma_busy_cycles_per_launch"] = int(mfma[k])
        out[name]["mfma_flop_per_launch"] = int(64 * mfma[k])
        step_mfma += 64 * mfma[k] * nm[k] / sq_steps
out["_step"] = {"hbm_bytes_per_step": int(step_bytes), "by_kernel": step_k, "steps_counted": steps}
if mfma:
    out["_step"]["mfma_flop_per_step"] = int(step_mfma)
# identity of the kernel sources this pass was taken with: bench.py reports `roofline.traffic` only while they match
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "socialways_amd", "csrc", "*.h*"))):
    if os.path.basename(f) not in ("sw_wide.hip", "sw_comm.hip"):       # as bench.kernel_src_sha16: in no measured step
        h.update(open(f, "rb").read())
commit = os.environ.get("SW_COMMIT")          # the GPU box has no .git: tools/collect_profiles.sh is given the commit
if not commit:
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = "unknown"
out["_meta"] = {"kernel_src_sha16": h.hexdigest()[:16], "commit": commit,
                "units": "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_launch = (2 FETCH + WRITE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["_step"], indent=1))
