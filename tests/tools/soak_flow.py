"""Soak run of the dataflow elimination (rlap_flow.hip): the random cases of soak.py with o_v = "random" forced through the
dataflow kernel (RLAP_FLOW=1, also for batches), a random workgroup shape and number of waves in flight, schedule jitter and
poisoned memory -- every result against the oracle.  usage: soak_flow.py SECONDS [SEED]"""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
sys.path.insert(0, os.path.join(_ROOT, "tests", "tools"))
import numpy as np
import soak


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    os.environ["RLAP_FLOW"] = "1"
    t_end = time.time() + budget
    n_cases = n_graphs = 0
    while time.time() < t_end:
        c = soak.draw(rs)
        c["o_v"] = "random"
        c["from_edges"] = False
        shape = str(rs.choice(["1", "2", "3"]))
        waves = str(rs.choice(["", "", "1", "5", "64", "700"]))
        os.environ["RLAP_FLOW_SHAPE"] = shape
        if waves:
            os.environ["RLAP_FLOW_WAVES"] = waves
        else:
            os.environ.pop("RLAP_FLOW_WAVES", None)
        if waves == "1" and c["n"] * c["G"] > 30000:     # one wave = the sequential order itself: small cases only
            os.environ["RLAP_FLOW_WAVES"] = "5"
        if os.environ.get("SOAK_VERBOSE"):
            print("case", n_cases, soak.describe(c), "flow shape", shape, "waves", os.environ.get("RLAP_FLOW_WAVES", "auto"), flush=True)
        bad = soak.run_case(c)
        if bad:
            print("MISMATCH case", n_cases, soak.describe(c), "flow shape", shape, "waves", waves or "auto", bad, flush=True)
            sys.exit(1)
        n_graphs += len(c["check"])
        n_cases += 1
        if n_cases % 10 == 0:
            print(f"[{n_cases} cases, {n_graphs} graphs checked] last: {soak.describe(c)} flow shape {shape} waves {waves or 'auto'}", flush=True)
    from rlap_amd import ops
    ops.debug_set_jitter(0)
    ops.debug_set_poison(-1)
    print(f"flow soak ok: {n_cases} cases, {n_graphs} graphs bit-exact against the oracle in {budget:.0f} s")


if __name__ == "__main__":
    main()
