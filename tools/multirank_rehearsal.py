"""bench.py --gpus N rehearsed on the 1-GPU box: N ranks under torch.distributed, ALL on device 0 (FIKSI_BENCH_DEVICE=0).
A code-path rehearsal of the N > 1 bench (rank seeds / shards, process-group init, barrier, max-over-ranks time, summed
counters), NOT a scaling number: the ranks share one GPU. Backend nccl (= RCCL) first; RCCL refuses communicators with two
ranks on one device on most builds — the refusal is recorded and the run repeated with gloo reductions
(FIKSI_BENCH_BACKEND=gloo), which exercises everything but the RCCL all-reduce itself; N = 1 with FIKSI_BENCH_FORCE_DIST=1
runs init / barrier / all-reduce on device tensors over RCCL for real. The last runs repeat two of them the way a launcher that
masks devices per rank would start them: HIP_VISIBLE_DEVICES=0 for every rank and NO FIKSI_BENCH_DEVICE, so that LOCAL_RANK 1 ... 3
meet one visible device and bench.py's masked-device branch (local_rank % devices visible) picks it.
    python3 tools/multirank_rehearsal.py > profiles/round5_multirank_rehearsal.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, scaling, backend, port, masked=False):
    env = dict(os.environ, FIKSI_BENCH_FORCE_DIST="1", FIKSI_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if masked:
        env["HIP_VISIBLE_DEVICES"] = "0"
        env.pop("FIKSI_BENCH_DEVICE", None)
    else:
        env["FIKSI_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "10", "--warmup", "2", "--quick",
           "--no-cpu-baseline", "--scaling", scaling]
    try:
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return None, "timed out after 600 s"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode or not lines:
        lines = [ln.strip() for ln in (p.stderr or "").splitlines() if ln.strip()]
        hits = [ln for ln in lines if any(w in ln for w in ("NCCL", "nccl", "Duplicate", "Error:", "error:"))]
        return None, " | ".join((hits or lines)[-4:])[-800:]
    d = json.loads(lines[0])
    return {"n_gpus": d["n_gpus"], "scaling": d["scaling"], "backend": backend, "value": d["value"], "unit": d["unit"],
            "ms_per_step": d["ms_per_step"], "steps": d["steps"], "converged_fraction": d["converged_fraction"],
            "global_systems": d["config"]["global_systems"], "systems_per_rank": d["config"]["systems_per_gpu"],
            "per_rank_ms": d.get("per_rank_ms"), "slowest_rank": d.get("slowest_rank"), "per_rank_systems": d.get("per_rank_systems"),
            "devices_masked_per_rank": bool(masked)}, None


def main():
    out = {"what": __doc__.split("\n    python3")[0].replace("\n", " "), "runs": [], "refusals": []}
    port = 29600
    for n, scaling in ((1, "weak"), (2, "weak"), (4, "weak"), (2, "strong"), (4, "strong")):
        for backend in ("nccl", "gloo"):
            port += 1
            r, err = run(n, scaling, backend, port)
            if r is not None:
                out["runs"].append(r)
                break
            out["refusals"].append({"n_gpus": n, "scaling": scaling, "backend": backend, "error_tail": err})
    for n, scaling in ((2, "weak"), (4, "strong")):  # HIP_VISIBLE_DEVICES=0 per rank: the masked-device branch
        port += 1
        r, err = run(n, scaling, "gloo", port, masked=True)
        if r is not None:
            out["runs"].append(r)
        else:
            out["refusals"].append({"n_gpus": n, "scaling": scaling, "backend": "gloo", "devices_masked_per_rank": True, "error_tail": err})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
