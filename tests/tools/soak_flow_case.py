"""Re-run ONE case of soak_flow.py (argv: SEED CASE_INDEX [jitter override]) -- the random draws are replayed up to that case."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, os.path.join(_ROOT, "tests", "tools"))
import numpy as np
import soak
seed, want = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(seed)
os.environ["RLAP_FLOW"] = "1"
for i in range(want + 1):
    c = soak.draw(rs)
    c["o_v"] = "random"; c["from_edges"] = False
    shape = str(rs.choice(["1", "2", "3"]))
    waves = str(rs.choice(["", "", "1", "5", "64", "700"]))
    if i < want:
        # the draws inside run_case are the case's own (seeded from c): nothing of rs is consumed there
        continue
    if "RLAP_FLOW_SHAPE" not in os.environ: os.environ["RLAP_FLOW_SHAPE"] = shape
    if waves: os.environ["RLAP_FLOW_WAVES"] = waves
    if waves == "1" and c["n"] * c["G"] > 30000: os.environ["RLAP_FLOW_WAVES"] = "5"
    if len(sys.argv) > 3: c["jitter"] = int(sys.argv[3])
    print("case", i, soak.describe(c), "flow shape", os.environ["RLAP_FLOW_SHAPE"], "waves", os.environ.get("RLAP_FLOW_WAVES", "auto"), flush=True)
    bad = soak.run_case(c)
    print("result:", bad or "bit-exact", flush=True)
