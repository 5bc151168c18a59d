"""Writes tests/golden/net_plan_*.txt: the canonical launch sequence of the planned executor (csrc/net.hip, trace mode) for
the benchmarked networks.  The sequences committed with this script were checked launch for launch - entry point, geometry, data
flow, counter-stream offsets - against the round-2 Python executor they replace (which ran the GPU parity suite green), before
that executor was deleted; regenerate only for a deliberate change of the plan.

    python tests/golden/make_net_plan_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import plan_trace as T  # noqa: E402

CASES = [("G32up-c", 8), ("G32up", 16), ("D32_st3", 8), ("G32up-c@64", 4), ("D32_st3@64", 4)]


def text(which, N):
    # one stream, and the localisation nets as separate modules (the configuration the deleted executor was compared in; the fused
    # localisation launches of csrc/locnet.hip came later and are covered by structure tests + the GPU parity suite)
    # ... and the weight gradients in line on the one stream (option wgrad_stream, round 4, moves them beside the data-gradient chain:
    # same launches, other stream - tests/test_net_plan.py::test_weight_gradients_run_beside_the_data_gradient_chain)
    r = T.trace(which, N, options=[("overlap_groups", 0), ("fuse_locnet", 0), ("pack_overlap", 0), ("head_fuse", 0), ("wgrad_stream", 0)])
    out = [f"# {which} batch {N}: draws {r['draws']}"]
    for phase in ("forward", "backward", "updateGradInput"):
        out.append(f"## {phase}")
        out += T.canon(r[phase])
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    for which, N in CASES:
        fn = os.path.join(HERE, "net_plan_%s_N%d.txt" % (which.replace("@", "_at_"), N))
        open(fn, "w").write(text(which, N))
        print(fn)
