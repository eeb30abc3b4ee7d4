#!/usr/bin/env python3
"""Run bench.py under several env settings / sizes and print one summary line each (GPU box)."""
import json, os, subprocess, sys
def run(env, args):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "bench.py", "--no-extras", "--no-cpu-baseline"] + args, capture_output=True, text=True, env=e)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        return d
    except Exception:
        print("FAILED", env, args, out.stderr[-500:])
        return None
if __name__ == "__main__":
    variants = [dict(kv.split("=") for kv in v.split("+")) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [{}]
    sizes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["20", "24"]
    for v in variants:
        for L in sizes:
            d = run(v, ["--logn", L, "--steps", "20"])
            if d:
                print(v, "logn", L, "Gel/s %.3f" % (d["value"] / 1e9), "ms/step %.4f" % d["ms_per_step"],
                      "roofline %.1f GB/s" % d["roofline"]["achieved"], d["check"]["roundtrip_ok"], flush=True)
