"""The planner below the C ABI (csrc/net.hip, cg_net_*), tested WITHOUT a GPU: in trace mode every launch goes to a recording
stub generated from include/catgan.h, so a whole forward / backward of G32up-c / G32up / D32_st3 (models.lua:138-160,196-228,
640-711,814-906) leaves its launch sequence as text.  What is pinned here:
  * the sequence itself (tests/golden/net_plan_*.txt: entry points, geometry, data flow, counter-stream offsets) - the fixtures
    were checked launch for launch against the round-2 Python executor before it was deleted (make_net_plan_golden.py);
  * the structure of the plan: fused segments, grouped / stacked / shared launches of D's three identical branches, the other
    branch group on a side stream between fork / join events, deferred reductions flushed once;
  * data flow: nothing reads a buffer no earlier launch (or the host) wrote;
  * the dropout draws follow the oracle's module-after-module order;
  * data parallelism: sync-BN sums exchanged between statistics and normalisation, gradient buckets started as their layers finish.
The arithmetic of every launch is the GPU suite's business (tests/test_gpu_parity*.py run the same plans on the MI355X)."""
import os
import re
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import plan_trace as T  # noqa: E402
from make_net_plan_golden import CASES, text  # noqa: E402
from oracle import oracle as O  # noqa: E402


@pytest.mark.parametrize("which,N", CASES, ids=[f"{w}-N{n}" for w, n in CASES])
def test_plan_equals_the_golden_launch_sequence(which, N):
    fn = os.path.join(HERE, "golden", "net_plan_%s_N%d.txt" % (which.replace("@", "_at_"), N))
    want = open(fn).read().split("\n")
    got = text(which, N).split("\n")
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, f"{which} N={N}: line {i} differs\n  plan  : {a[:300]}\n  golden: {b[:300]}"
    assert len(got) == len(want)


def test_segments_and_lockstep_branches_of_the_discriminator():
    r = T.trace("D32_st3", 128)
    f = r["forward"]
    names = [c[0] for c in T.calls(f)]
    # the localisation branches (models.lua:842-860 + :877-878) run as ONE launch each: the first transformer's, and the three branch
    # transformers' together (ngroups 3 on the shared trunk output)
    loc = T.calls(f, "cg_locnet_forward")
    assert [(a["ngroups"], a["x_shared"], a["S"], a["Cin"], a["P"]) for _, a in loc] == [("i:1", "i:1", "i:16", "i:3", "i:1"),
                                                                                       ("i:3", "i:1", "i:8", "i:64", "i:4")]
    assert not [n for n in names if n.startswith(("cg_affine_", "cg_avgpool2", "cg_leakyrelu"))]
    # three identical transformer branches: one shared-image sampler launch, convolutions grouped (ngroups 3), pooling stacked
    ex = T.calls(f, "cg_conv2d_forward_ex")
    assert sum(1 for _, a in ex if a["ngroups"] == "i:3") == 1          # branch conv2 + PReLU
    grouped = T.calls(f, "cg_conv2d_forward_grouped")
    assert len(grouped) == 1 and all(a["ngroups"] == "i:3" for _, a in grouped)       # branch conv1 (PReLU goes with the pooling)
    assert len(T.calls(f, "cg_bilinear_sampler_forward_shared")) == 1 and len(T.calls(f, "cg_bilinear_sampler_forward")) == 1
    pools = T.calls(f, "cg_act_pool2_mask_forward")
    assert sorted(a["ngroups"] for _, a in pools) == ["i:1", "i:1", "i:3"]
    # nn.Concat -> nn.SpatialDropout in one launch, mask drawn inside; [nn.Dropout, nn.Linear(256, 1), nn.Sigmoid] in one launch
    assert names.count("cg_concat_channels_dropout") == 1 and "cg_concat_channels" not in names and "cg_copy_channels" not in names
    assert names[-1] == "cg_drop_linear_sigmoid_forward" and "cg_sigmoid_forward" not in names and "cg_mask_mul" not in names
    head = T.calls(f, "cg_drop_linear_sigmoid_forward")[0][1]
    assert (head["N"], head["F"], head["O"]) == ("i:128", "i:256", "i:1")
    # nn.View -> nn.Linear on the NHWC map: no layout pass in front of the 20480 -> 256 layer
    assert "cg_nhwc_to_nchw" not in names
    # side stream: the two-convolution branch runs on s1 between fork and join
    s1 = [c for c in T.calls(f) if c[1]["stream"] == "s1"]
    assert {c[0] for c in s1} == {"cg_conv2d_forward", "cg_act_pool2_mask_forward", "cg_rng_bernoulli_dev", "cg_conv2d_forward_ex"}
    ev = [l for l in f if l.startswith("event|")]
    assert ev == ["event|record|fork|s0", "event|wait|fork|s1", "event|record|join1|s1", "event|wait|join1|s0"]
    assert f.index("event|wait|join1|s0") < next(i for i, l in enumerate(f) if "cg_concat_channels_dropout" in l)
    # backward: weight-gradient reductions deferred, one flush per stream, grads of the three branches in grouped launches
    b = r["backward"]
    bn = [c[0] for c in T.calls(b)]
    assert bn[0] == "cg_drop_linear_sigmoid_backward" and bn.count("cg_split_channels_masked") == 1 and "cg_split_channels" not in bn
    assert "cg_sigmoid_backward" not in bn and "cg_mask_mul" not in bn
    # (the weight-gradient launches and their flushes sit on s4 / s5 beside the two branch groups' streams s0 / s1: next test)
    assert len(T.calls(b, "cg_conv2d_wgrad_flush")) == 2 and not T.calls(b, "cg_conv2d_wgrad")
    assert {a["stream"] for _, a in T.calls(b, "cg_conv2d_wgrad_flush")} == {"s4", "s5"}
    assert b[-6:] == [b[-6], "event|record|wgjoin0|s4", "event|wait|wgjoin0|s0", b[-3], "event|record|wgjoin1|s5", "event|wait|wgjoin1|s0"]
    assert b[-6].startswith("call|cg_conv2d_wgrad_flush|s4") and b[-3].startswith("call|cg_conv2d_wgrad_flush|s5")
    assert len(T.calls(b, "cg_bilinear_sampler_backward_shared")) == 1
    assert len(T.calls(b, "cg_prelu_backward_grouped")) == 1      # the PReLUs behind the branches' second convolutions (the first ones are inside act_pool)
    lb = T.calls(b, "cg_locnet_backward")
    assert [a["ngroups"] for _, a in lb] == ["i:3", "i:1"]
    # each localisation backward is followed by the four weight gradients of its layers (conv1, conv2, linear1, linear2)
    # (on the weight-gradient stream of the branch group's stream, forked right behind the launch that produces their gradOutputs)
    for i, l in enumerate(b):
        if "cg_locnet_backward" in l:
            assert b[i + 1].startswith("event|record|wgfork") and b[i + 2].startswith("event|wait|wgfork")
            assert all("cg_conv2d_wgrad_grouped_deferred" in x for x in b[i + 3:i + 7])
    # updateGradInput only (fevalG_on_D's pass through D, adversarial.lua:192-193): no weight gradient of any kind
    u = r["updateGradInput"]
    assert not [c for c in T.calls(u) if "wgrad" in c[0]]
    hb = T.calls(u, "cg_drop_linear_sigmoid_backward")[0][1]
    assert hb["gw"] == "n" and hb["gb"] == "n"
    assert r["stats"]["launches_forward"] <= 23 and len(T.calls(b)) <= 44 and r["stats"]["launches_backward"] <= 66   # ops: launches + stream events
    # head_fuse 0: the modules of the head one by one, every draw at the same position of the counter stream
    rh = T.trace("D32_st3", 128, options=[("head_fuse", 0)])
    assert rh["draws"] == r["draws"]
    hn = [c[0] for c in T.calls(rh["forward"])]
    assert hn.count("cg_concat_channels") == 1 and hn[-1] == "cg_sigmoid_forward" and hn.count("cg_mask_mul") == 2
    off = lambda calls, name: [a["offset"] for _, a in T.calls(calls, name)]
    assert off(f, "cg_concat_channels_dropout") + off(f, "cg_drop_linear_sigmoid_forward") == off(rh["forward"], "cg_rng_bernoulli_dev")[-2:]
    # the same network with the localisation nets as separate modules (cg_net_set_option fuse_locnet 0): ~1.5x the launches
    r0 = T.trace("D32_st3", 128, options=[("fuse_locnet", 0)])
    assert r0["stats"]["launches_forward"] >= 35 and not T.calls(r0["forward"], "cg_locnet_forward")


def test_generator_plan_uses_epilogue_statistics_and_winograd_at_the_benchmarked_batch():
    r = T.trace("G32up-c", 128)
    f = [c[0] for c in T.calls(r["forward"])]
    # conv -> BN -> PReLU: statistics partials from the GEMM epilogue (no separate pass over the convolution output)
    assert f.count("cg_bn_stats_finalize") == 3 and "cg_bn_stats" not in f and f.count("cg_bn_act_forward") == 3
    assert f.count("cg_conv2d_ups2_wino_forward_stats") == 1      # the 5x5 layer's phases in Winograd F(2x2,3x3)
    assert "cg_upsample2x_forward" not in f and "cg_prelu_forward" not in f
    b = [c[0] for c in T.calls(r["backward"])]
    assert b.count("cg_conv2d_ups2_wino_dgrad") == 1 and b.count("cg_conv2d_ups2_wino_wgrad") == 1 and b.count("cg_bn_act_backward_stats") == 3
    # a Winograd data gradient whose unsplit launch is below one workgroup per CU runs in K slices over blockIdx.z + a fixed-order sum (option
    # wino_dsplit): G32up's 128 -> 256 layer at batch 256 (64 workgroups), not its 256 -> 128 layer (512) nor G32up-c's (256, above)
    b3 = [c[0] for c in T.calls(T.trace("G32up", 256)["backward"])]
    assert b3.count("cg_conv2d_ups2_wino_dgrad_split") == 1 and b3.count("cg_conv2d_ups2_wino_dgrad") == 1
    b3u = [c[0] for c in T.calls(T.trace("G32up", 256, options=[("wino_dsplit", 0)])["backward"])]
    assert b3u.count("cg_conv2d_ups2_wino_dgrad") == 2 and "cg_conv2d_ups2_wino_dgrad_split" not in b3u
    # the 512 -> 256 3x3 layer behind the 8x8 -> 16x16 upsampling: forward and data gradient in F(2x2,2x2), the weight gradient phase-folded
    assert f.count("cg_conv2d_ups2_wino22_forward_stats") == 1 and b.count("cg_conv2d_ups2_wino22_dgrad") == 1
    assert b.count("cg_conv2d_dgrad_ups2") == 1 and b.count("cg_bn_act_backward") == 3 and b[-1] == "cg_conv2d_wgrad_flush"
    r22 = T.trace("G32up-c", 128, options=[("winograd22", 1)])     # bit 0 only: the forward alone
    b22 = [c[0] for c in T.calls(r22["backward"])]
    assert b22.count("cg_conv2d_dgrad_ups2") == 2
    r20 = T.trace("G32up-c", 128, options=[("winograd22", 0)])
    assert "cg_conv2d_ups2_wino22_forward_stats" not in [c[0] for c in T.calls(r20["forward"])]
    assert r["backward"][-2:] == ["event|record|wgjoin0|s4", "event|wait|wgjoin0|s0"]
    small = [c[0] for c in T.calls(T.trace("G32up-c", 8)["forward"])]
    assert "cg_conv2d_ups2_wino_forward_stats" not in small and "cg_conv2d_ups2_wino22_forward_stats" not in small   # below 2048 tiles the direct phase kernels run


PTR_ARG = {"void*", "const void*", "float*", "const float*", "double*", "const double*", "int32_t*", "const int32_t*", "uint64_t*",
           "const uint64_t*"}


@pytest.mark.parametrize("which,N", [("D32_st3", 16), ("G32up-c", 16), ("G32up", 8)])
def test_no_launch_reads_what_nothing_wrote(which, N):
    """Data-flow check over forward + backward: a `const T*` argument is an input; every input region must have been written by an
    earlier launch (a non-const pointer argument) or belong to the host (parameters, input, gradOutput)."""
    r = T.trace(which, N)
    P = T.protos()
    pn = r["net"]
    host_regions = set()
    # regions registered by the host come after the plan's own allocations in the list: find them through a marker call
    lines = r["first_forward"] + r["backward"]      # the first pass includes the weight packing
    written = set()

    def reg(tok):
        m = re.match(r"r(\d+)\+", tok)
        return m.group(1) if m else None

    # the plan's scratch is allocated zeroed, the statistics buffers start as zeros: reading zero-initialised sums is legal;
    # everything else must follow a write.  Collect the host-owned regions from the parameter / input pointers.
    for l in lines:
        f = l.split("|")
        if f[0] != "call":
            continue
        for (t, an), v in zip(P[f[1]][1], f[2:]):
            if an in ("wpk", "bias", "gamma", "beta", "alpha", "w_canonical", "gw_canonical", "gb", "ggamma", "gbeta", "galpha", "running_mean",
                      "running_var") and v not in ("n",):
                for tok in (v.split(":", 2)[2].split(",") if v.startswith("a:") else [v]):
                    if reg(tok):
                        host_regions.add(reg(tok))
    first = r["forward"][0].split("|")
    bad = []
    inputs_seen = set()
    for l in lines:
        f = l.split("|")
        if f[0] != "call":
            continue
        for (t, an), v in zip(P[f[1]][1], f[2:]):
            if t in ("const float* const*", "float* const*"):
                toks, const = ([] if v == "n" else v.split(":", 2)[2].split(",")), t.startswith("const")
            elif t in PTR_ARG and an != "stream":
                toks, const = [v], t.startswith("const")
            else:
                continue
            for tok in toks:
                k = reg(tok)
                if k is None or an in ("ws", "base"):
                    continue
                if const:
                    if k not in written and k not in host_regions:
                        inputs_seen.add(k)
                        bad.append((f[1], an, tok))
                else:
                    written.add(k)
    # the only never-written inputs are the caller's x and gradOutput (two regions)
    assert len({reg(b[2]) for b in bad}) <= 2, bad[:5]


def test_dropout_draws_follow_the_oracle_order():
    """The masks of a planned pass sit at the counter-stream offsets a module-after-module walk draws them at (the oracle's order):
    same total, and the per-launch offsets partition [offset, offset + draws) without gaps or overlaps."""
    N = 8
    r = T.trace("D32_st3", N, rng_offset=5000)
    rng = O.RNG(3)
    Do = O.create_D32_st3(3, 32, rng)
    o0 = rng.offset
    Do.forward(np.zeros((N, 3, 32, 32), np.float32))
    assert r["draws"] == rng.offset - o0
    spans = []
    for name, a in T.calls(r["forward"]):
        if name == "cg_rng_bernoulli_dev":
            spans.append((int(a["offset"][2:]), int(a["n"][2:])))
        elif name == "cg_rng_bernoulli_dev_grouped":
            n, G = int(a["n_per_group"][2:]), int(a["ngroups"][2:])
            spans += [(int(a[f"off{g}"][2:]), n) for g in range(G)]
        elif name == "cg_concat_channels_dropout":      # mask drawn inside the launch: [N][sum of the branches' channels]
            spans.append((int(a["offset"][2:]), int(a["N"][2:]) * sum(int(c) for c in a["C"].split(":", 1)[1].split(","))))
        elif name == "cg_drop_linear_sigmoid_forward":  # [N][F]
            spans.append((int(a["offset"][2:]), int(a["N"][2:]) * int(a["F"][2:])))
    spans.sort()
    pos = 5000
    for off, n in spans:
        assert off == pos, (off, pos)
        pos += n
    assert pos == 5000 + r["draws"]
    # the three branches' first masks (SpatialDropout(0.2) behind the max pool) are 64 channels x N samples each, in branch order
    g = [a for n_, a in T.calls(r["forward"], "cg_rng_bernoulli_dev_grouped")][0]
    offs = [int(g[f"off{i}"][2:]) for i in range(3)]
    assert offs[1] - offs[0] == offs[2] - offs[1] == N * 64          # each branch draws one [N,64] mask (models.lua:658)


def test_data_parallel_exchanges_sit_inside_the_plan():
    """world 2 (trace mode: the exchanges appear as hook lines): sync-BN sums between statistics and normalisation, forward and
    backward; G's flat gradient in four buckets (one per convolution / linear layer, with the BN / PReLU parameters behind it; the small
    last convolution rides with the layer in front), each started behind its layer's weight gradient, the deferred reductions flushed first."""
    r = T.trace("G32up-c", 128, dp=dict(world=2, buckets=True))
    f = r["forward"]
    hooks = [i for i, l in enumerate(f) if l.startswith("hook|allreduce_sum")]
    assert len(hooks) == 3
    for i in hooks:
        assert "cg_bn_stats_finalize" in f[i - 1] and "cg_bn_act_forward" in f[i + 1]
        cnt = [a for n_, a in T.calls([f[i + 1]])][0]["count"]
        assert float.fromhex(cnt[2:]) in (2.0 * 128 * 64, 2.0 * 128 * 256, 2.0 * 128 * 1024)      # global-batch count
    b = r["backward"]
    assert sum(1 for l in b if l.startswith("hook|allreduce_sum")) == 3
    buckets = [(i, l.split("|")) for i, l in enumerate(b) if l.startswith("hook|bucket_start")]
    assert len(buckets) == 4
    counts = [int(t[3]) for _, t in buckets]
    lin = 100 * 8192 + 8192 + 1
    c1 = 512 * 512 * 9 + 512 + 2 * 512 + 1
    c2 = 512 * 256 * 9 + 256 + 2 * 256 + 1
    c3 = 256 * 128 * 25 + 128 + 2 * 128 + 1
    c4 = 128 * 3 * 9 + 3
    # reverse layer order, the whole vector; the last convolution's 3 459 values ride with the layer in front (buckets below 256 KB merge)
    assert counts == [c4 + c3, c2, c1, lin] and sum(counts) == 5191687
    offs = [int(t[2].split("+")[1]) // 4 for _, t in buckets]
    assert offs == [lin + c1 + c2, lin + c1, lin, 0]
    for i, _ in buckets[:-1]:
        # complete gradients: the deferred reductions flushed, or the layer's own immediate (Winograd-domain) weight gradient
        # (on the weight-gradient stream s4, which first takes up everything s0 has issued: BN / PReLU gradients of the bucket)
        prev = [l for l in b[:i] if not l.startswith("event|")][-1]
        assert "cg_conv2d_wgrad_flush|s4" in prev or "cg_conv2d_ups2_wino_wgrad|s4" in prev or prev.startswith("hook|")
        assert b[i - 1] == "event|wait|wgfork0|s4" or "cg_conv2d_wgrad_flush|s4" in b[i - 1]
    for (i0, _), (i1, _) in zip(buckets, buckets[1:]):                                   # the next layer's backward runs under the bucket
        assert sum(1 for l in b[i0:i1] if l.startswith("call|")) >= 3
    # round 6: while a batch-norm layer in front still has its exchange to make, a bucket is held back and started right BEHIND that
    # exchange (collectives of the two communicators are ordered on the device: started at once, the bucket - which sits behind its
    # layer's whole weight gradient - would stall the next layer's sync-BN sums, the head of the data-gradient chain)
    syncs = [i for i, l in enumerate(b) if l.startswith("hook|allreduce_sum")]
    for (ib, _), isync in zip(buckets[:2], syncs[1:]):
        between = [l for l in b[isync + 1:ib] if not l.startswith("event|")]
        assert isync < ib and all("cg_conv2d_wgrad_flush|s4" in l for l in between), (isync, ib, between)
        assert "cg_bn_act_backward|s0" in [l for l in b[ib + 1:] if l.startswith("call|")][0]     # the normalisation's backward follows
    assert all(ib > syncs[-1] for ib, _ in buckets[2:])          # no exchange left in front: these start where their layer ends


def test_discriminator_gradient_travels_in_buckets_too():
    """Round 6: D32_st3's flat gradient in buckets started from inside cg_net_backward (adversarial.py sets _bucket_overlap on D as well):
    nn.Linear(20480, 256) - 79 % of the vector - is the first large bucket the backward completes and travels under everything behind it.
    The buckets tile the whole vector in reverse order, and a bucket that holds the nn.Concat's branches starts on a stream that has
    waited for the branch groups' weight-gradient streams (s5..s7), which nothing else joins before the end of the pass."""
    r = T.trace("D32_st3", 128, dp=dict(world=2, buckets=True))
    b = r["backward"]
    buckets = [(i, l.split("|")) for i, l in enumerate(b) if l.startswith("hook|bucket_start")]
    assert len(buckets) == 3                                     # [head + Linear(20480, 256)], [nn.Concat], [the first layers]
    counts = [int(t[3]) for _, t in buckets]
    offs = [int(t[2].split("+")[1]) // 4 for _, t in buckets]
    assert sum(counts) == 6664777 and offs[-1] == 0
    for (o0, c0), o1 in zip(zip(offs[1:], counts[1:]), offs):
        assert o0 + c0 == o1                                     # contiguous, last layer first
    assert max(counts) >= 20480 * 256 and counts.index(max(counts)) == 0          # the head's bucket is the first one started
    big = buckets[counts.index(max(counts))][0]
    assert sum(1 for l in b[big:] if l.startswith("call|")) >= 30                 # ... with the rest of the backward behind it
    # the bucket that covers the branches: every used branch weight-gradient stream is recorded and waited for before it starts
    used = sorted({l.split("|")[2] for l in b if l.startswith("call|") and l.split("|")[2] in ("s5", "s6", "s7")})
    assert used, "the branch groups' weight gradients run on their own streams"
    last_branch_call = max(i for i, l in enumerate(b) if l.startswith("call|") and l.split("|")[2] in used)
    nxt = min(i for i, _ in buckets if i > last_branch_call)
    for s_ in used:
        k = s_[1:]
        rec = [i for i, l in enumerate(b[:nxt]) if l.startswith(f"event|record|wgjoin{int(k) - 4}|")]
        assert rec and rec[-1] > last_branch_call and any(l.startswith(f"event|wait|wgjoin{int(k) - 4}|") for l in b[rec[-1]:nxt])


@pytest.mark.parametrize("which,N", [("D32_st3", 128), ("G32up-c", 128), ("G32up", 256), ("D32_st3@64", 64)])
def test_weight_gradients_run_beside_the_data_gradient_chain(which, N):
    """Option wgrad_stream (default on): Module:backward issues every accGradParameters launch - GEMM, Winograd-domain, the
    localisation nets' four, the deferred reductions - on stream 4 + s, forked from stream s where the layer's gradOutput is complete
    and joined into s0 once, at the end of the pass.  Nothing else changes: with the stream column and the fork / join events removed
    the plan IS the in-line plan (wgrad_stream 0, the configuration the golden sequences hold) up to the position of the launches
    that moved, and no data-gradient launch waits for a weight gradient."""
    r1 = T.trace(which, N)
    r0 = T.trace(which, N, options=[("wgrad_stream", 0)])
    b1, b0 = r1["backward"], r0["backward"]
    is_w = lambda l: l.startswith("call|cg_conv2d_wgrad") or l.startswith("call|cg_conv2d_ups2_wino_wgrad")
    # same launches, same arguments (workspace of the stream aside), same relative order within the weight gradients and within the rest
    def strip(l):      # entry point + every scalar argument; streams and buffer names dropped (holding a launch back moves the allocation
        f = l.split("|")   # order of its workspace, and with it the numbering of the regions)
        return "|".join([f[0], f[1]] + [t for t in f[3:] if t[:2] in ("i:", "f:", "I:") and not t.startswith("u:")])
    calls1 = [l for l in T.canon(b1) if l.startswith("call|")]
    calls0 = [l for l in T.canon(b0) if l.startswith("call|")]
    nonflush = lambda ls: [strip(l) for l in ls if "wgrad_flush" not in l]
    assert sorted(nonflush(calls1)) == sorted(nonflush(calls0))
    assert [strip(l) for l in calls1 if not is_w(l)] == [strip(l) for l in calls0 if not is_w(l)]
    # the weight gradients keep their order PER STREAM (s4 + k carries what stream k carried in line)
    on = lambda ls, st: nonflush([l for l in ls if is_w(l) and l.split("|")[2] == st])
    for k_ in (0, 1):
        assert on(T.canon(b1), "s%d" % (4 + k_)) == on(T.canon(b0), "s%d" % k_)
    # where they run
    w1 = [l.split("|")[2] for l in b1 if is_w(l)]
    assert w1 and set(w1) <= {"s4", "s5"} and not [l for l in b1 if l.startswith("call|") and not is_w(l) and l.split("|")[2] in ("s4", "s5")]
    assert all(l.split("|")[2] in ("s0", "s1") for l in b0 if is_w(l))
    # every weight-gradient launch on s(4+k) is behind a wait on the fork event recorded on s(k) AFTER the launch that produced its
    # gradOutput; the pass ends by joining the weight-gradient streams into s0
    last_fork = {}
    for l in b1:
        f = l.split("|")
        if f[0] == "event" and f[2].startswith("wgfork"):
            if f[1] == "record":
                last_fork[f[2]] = "recorded"
            else:
                assert last_fork.get(f[2]) == "recorded" and f[3] == "s%d" % (4 + int(f[2][-1]))
                last_fork[f[2]] = "waited"
    joins = [l for l in b1 if l.startswith("event|") and "wgjoin" in l]
    assert joins and b1[-1] == joins[-1] and all(j.endswith("|s0") for j in joins if "|wait|" in j)
    for k in {s[1] for s in set(w1)}:
        assert "event|record|wgjoin%d|s%s" % (int(k) - 4, k) in b1
    # updateGradInput (no weight gradients) is untouched
    assert T.canon(r1["updateGradInput"]) == T.canon(r0["updateGradInput"]) and not [l for l in r1["updateGradInput"] if "wgfork" in l or "wgjoin" in l]
    assert T.canon(r1["forward"]) == T.canon(r0["forward"])


def test_per_module_walk_is_still_the_protocol():
    """nn.planned = False (or a module used on its own): updateOutput / updateGradInput / accGradParameters module by module."""
    cg = T.cg_pkg()
    net, _, _ = T.build("G32up-c")
    assert type(net) is cg.nn.Sequential and not net._planned_last
    assert [type(m).__name__ for m in net.modules[:3]] == ["Linear", "PReLU", "View"]
    assert cg.nn.planned is True


def test_weight_packing_runs_beside_the_head_of_the_pass():
    """After a parameter update only the FIRST layer behind a folded upsampling needs its phase-summed weights at once; the others
    (and the Winograd-domain kernels) are re-packed on side stream 1, and the pass waits for them right in front of the first launch
    that reads them.  pack_overlap 0 keeps everything on the pass's own stream."""
    r = T.trace("G32up-c", 64)
    first = r["first_forward"]
    packs = [l.split("|") for l in first if l.startswith("call|cg_pack_conv_weight_ups2") or l.startswith("call|cg_conv2d_ups2_wino_pack")]
    assert [p[2] for p in packs] == ["s0", "s1", "s1", "s1"]
    side_outputs = set()
    for p in packs[1:]:
        side_outputs.update(t for t in (p[4:6] if "wino" not in p[1] else p[5:7]))
    wait = first.index("event|wait|packs|all")
    assert first.index("event|record|packs|s1") < wait
    readers = [i for i, l in enumerate(first) if l.startswith("call|cg_conv2d") and "pack" not in l and any(t in l.split("|") or any(t in a for a in l.split("|")) for t in side_outputs)]
    assert readers and min(readers) == wait + 1, (wait, readers[:3])
    # the second pass (nothing dirty) has no packing and no events
    assert not any("pack" in l for l in r["forward"])
    r0 = T.trace("G32up-c", 64, options=[("pack_overlap", 0)])
    assert all(l.split("|")[2] == "s0" for l in r0["first_forward"] if l.startswith("call|"))
    assert not any(l.startswith("event|") and "packs" in l for l in r0["first_forward"])
    strip = lambda ls: [l for l in ls if not (l.startswith("event|") and "packs" in l)]
    assert [l.replace("|s1|", "|s0|") for l in strip(first)] == r0["first_forward"]


def test_net_options_from_the_environment(monkeypatch):
    """CG_NET_OPTIONS="name=value,..." (round 5: ONE variable instead of eighteen): every net created while it is set takes the pairs
    through cg_net_set_option - the plan equals the one built with the same options through the ABI; an unknown name or a pair
    without '=' fails cg_net_create with a message, it is not ignored."""
    cg = T.cg_pkg()
    want = T.trace("G32up-c", 128, options=[("winograd", 0), ("wgrad_stream", 0)])
    monkeypatch.setenv("CG_NET_OPTIONS", "winograd=0,wgrad_stream=0")
    got = T.trace("G32up-c", 128)
    assert T.canon(got["forward"]) == T.canon(want["forward"]) and T.canon(got["backward"]) == T.canon(want["backward"])
    assert not any("wino" in l for l in got["forward"])
    monkeypatch.delenv("CG_NET_OPTIONS")
    assert any("wino" in l for l in T.trace("G32up-c", 128)["forward"])
    for bad in ("no_such_option=1", "winograd"):
        monkeypatch.setenv("CG_NET_OPTIONS", bad)
        with pytest.raises(cg._abi.CatganError, match="no_such_option|name=value"):
            T.trace("G32up-c", 128)
