"""adversarial.lua's training iteration on the engine (the hot path; adversarial.lua:27-292).

`State` carries what train.lua keeps in Lua globals (OPT, MODEL_G/D, CRITERION, PARAMETERS_*, OPTSTATE,
CONFUSION, ...; train.lua:51-220).  `train()` is adversarial.train(); `iteration()` is the body of its for-loop
(one D update + one G update, :51-275).  Batches, targets, noise and gradients-w.r.t.-images stay in HBM; the
reference's per-image host loops (:225-238) become one gather kernel + one device copy.

Differences from the reference, all inert on results:
  * fevalG_on_D calls MODEL_D:updateGradInput instead of :backward — the D weight gradients the reference
    accumulates there (:192) are zeroed by the next fevalD (:81) before anyone reads them
    (OPT.exact_reference_backward=True restores the wasted work);
  * penalty + clamp + Adam run as one fused kernel (OPT.fused_update=False restores the separate passes);
  * the rolling-accuracy gate (:144-166) only reads the confusion counts back when D_maxAcc <= 1 can trigger.
Under data parallelism the flat gradient is all-reduced (mean) right after backward, i.e. before penalty and
clamp — the order adversarial.lua:89-112 fixes.
"""
import math
import time

import os

import numpy as np
import torch

from . import nn, nn_utils, optim, parallel
from .tensor import Tensor, has_gpu, lib, rng, stream

DEFAULT_OPT = dict(  # train.lua:15-49
    batchSize=32, N_epoch=1000, G_L1=0.0, G_L2=0.0, D_L1=0.0, D_L2=1e-4, D_iterations=1, G_iterations=1,
    D_maxAcc=1.01, D_clamp=1.0, G_clamp=5.0, D_optmethod="adam", G_optmethod="adam", noiseDim=100, scale=32,
    seed=1, colorSpace="rgb", fused_update=True, exact_reference_backward=False, overlap_comm=True,
    # BOTH generator forwards of an iteration - the D-step's fake images on N/2 noise rows (adversarial.lua:232-233) and the G-step's pass
    # on N rows (:185) - read the same G parameters (G moves at :262 only), so the second one starts at the head of the iteration beside
    # the first instead of after D's update (round 5: 6.15 -> 5.95 ms; since round 6 ONE call below the ABI, cg_net_forward_pair, so a
    # LuaJIT host gets the same schedule from MODEL_G:forwardPair).  Result-neutral: the G-step's noise is drawn from the position it has
    # in the reference's order (behind the D-step's dropout masks), the second pass leaves the batch-norm running statistics alone and
    # the library moves them behind the first pass's update.  CG_CONCURRENT_G_BOTH=0 = the reference's order of calls.
    concurrent_g_both=os.environ.get("CG_CONCURRENT_G_BOTH", "1") != "0",
)


class TrainData:
    """A pool of real images resident in HBM ([P,C,H,W], NHWC) with the bits of the dataset table API the loop
    uses: size() and random row access (adversarial.lua:226-227)."""

    def __init__(self, pool):
        self.pool = nn.as_nhwc(nn.to_device(pool))

    def size(self):
        return self.pool.shape[0]


class State:
    Y_GENERATOR = 0
    Y_NOT_GENERATOR = 1
    CLASSES = ("0", "1")

    def __init__(self, OPT, MODEL_G, MODEL_D):
        self.OPT = dict(DEFAULT_OPT)
        self.OPT.update(OPT)
        self.MODEL_G, self.MODEL_D = MODEL_G, MODEL_D
        self.CRITERION = nn.BCECriterion()                                   # train.lua:181
        self.PARAMETERS_D, self.GRAD_PARAMETERS_D = MODEL_D.getParameters()   # train.lua:184
        self.PARAMETERS_G, self.GRAD_PARAMETERS_G = MODEL_G.getParameters()   # train.lua:185
        self.CONFUSION = optim.ConfusionMatrix(self.CLASSES)                  # train.lua:188
        o = self.OPT                                                          # train.lua:191-207
        self.OPTSTATE = {"adagrad": {"D": {"learningRate": 1e-3}, "G": {"learningRate": 1e-3 * 3}},
                         "adam": {"D": {}, "G": {}},
                         "sgd": {"D": {"learningRate": o.get("D_sgd_lr", 0.02), "momentum": o.get("D_sgd_momentum", 0)},
                                 "G": {"learningRate": o.get("G_sgd_lr", 0.02), "momentum": o.get("G_sgd_momentum", 0)}}}
        self.EPOCH = 1
        self.random = np.random.RandomState(self.OPT["seed"])  # math.randomseed(OPT.seed), train.lua:61
        self.accs = []
        self._cache = {}
        self.keep_outputs = False  # tests: snapshot D's output before the G-step reuses the buffer
        self.device_rng = False    # draw the real-batch indices on the device (required under hipGraph replay)
        self._d_draws = {}         # batch -> counter-stream draws of MODEL_D:forward (learned in the first iteration)


def mean(t):
    """adversarial.mean (adversarial.lua:12-24)."""
    v = [x for x in t if isinstance(x, (int, float))]
    return sum(v) / len(v)


def _buffers(S, thisBatchSize, dims):
    key = (thisBatchSize, tuple(dims))
    b = S._cache.get(key)
    if b is None:
        half = thisBatchSize // 2
        b = dict(
            inputs=Tensor.zeros((thisBatchSize,) + tuple(dims), "nhwc"),
            targets_D=Tensor.from_numpy(np.concatenate([np.full(half, S.Y_NOT_GENERATOR, np.float32),
                                                        np.full(half, S.Y_GENERATOR, np.float32)])),
            targets_G=Tensor.from_numpy(np.full(thisBatchSize, S.Y_NOT_GENERATOR, np.float32)),
            idx=torch.zeros(half, dtype=torch.int32, device=S.PARAMETERS_D.t.device),
        )
        S._cache[key] = b
    return b


def iteration(S, trainData, thisBatchSize=None, maxAccuracyD=1.01, accsInterval=20, real_idx=None, noise_D=None,
              noise_G=None):
    """One pass of the loop body adversarial.lua:51-275.  real_idx / noise_D / noise_G inject the random draws
    (parity tests); by default they come from the seeded host RNG / the device counter stream."""
    OPT = S.OPT
    N = thisBatchSize or OPT["batchSize"]
    assert N >= 4 and N % 2 == 0  # adversarial.lua:65-68
    dims = trainData.pool.shape[1:]
    buf = _buffers(S, N, dims)
    inputs, half = buf["inputs"], N // 2
    rowlen = int(np.prod(dims))
    st = {"doTrainD": True}
    # data parallelism: G's gradient travels in per-layer buckets started from inside the planned backward
    S.MODEL_G._bucket_overlap = bool(parallel.world_size() > 1 and OPT.get("overlap_comm", True) and _bucketable(S.MODEL_G) and nn.planned)
    # ... and so does D's (round 6): Linear(20480, 256) is 79 % of D's gradient and the FIRST bucket its backward completes, so it travels
    # under the whole of D's backward instead of after it (with both generator forwards at the head of the iteration nothing else is left
    # to hide it under).  Not with exact_reference_backward, whose G-step would start exchanges of a D gradient nobody reads.
    S.MODEL_D._bucket_overlap = bool(S.MODEL_G._bucket_overlap and _bucketable(S.MODEL_D) and not OPT["exact_reference_backward"])

    # ------------------------------------------------------------------ fevalD (adversarial.lua:72-167)
    def fevalD(x):
        if x is not S.PARAMETERS_D:
            S.PARAMETERS_D.copy(x)
        S.GRAD_PARAMETERS_D.zero()
        targets = buf["targets_D"]
        outputs = S.MODEL_D.forward(inputs)
        pd = getattr(S.MODEL_D, "_pnet", None)
        if pd and pd[1] is not None and getattr(S.MODEL_D, "_planned_last", False):
            S._d_draws[N] = pd[1].last_draws     # counter-stream draws of D's forward at this batch: static per plan
        f = S.CRITERION.forward(outputs, targets)
        df_do = S.CRITERION.backward(outputs, targets)
        S.MODEL_D.backward(inputs, df_do)   # planned: weight-gradient reductions deferred and flushed inside cg_net_backward
        if getattr(S.MODEL_D, "_bucket_overlap", False) and S.MODEL_D._planned_last and S.MODEL_D._pnet[1].finish_buckets():
            st["pendingD"] = _Done()        # the buckets travelled under the backward and are joined
        elif st.get("overlap"):  # start the xGMI all-reduce now, finish it after the G-step's generator forward
            st["pendingD"] = parallel.allreduce_mean_async(S.GRAD_PARAMETERS_D.t)
        else:
            parallel.allreduce_mean_(S.GRAD_PARAMETERS_D.t)
        if not OPT["fused_update"]:
            if OPT["D_L1"] != 0 or OPT["D_L2"] != 0:
                f = float(f) + OPT["D_L1"] * S.PARAMETERS_D.norm(1) + OPT["D_L2"] * S.PARAMETERS_D.norm(2) ** 2 / 2
                if OPT["D_L1"] != 0:
                    lib().axpy_sign(stream(), OPT["D_L1"], S.PARAMETERS_D.ptr, S.GRAD_PARAMETERS_D.ptr,
                                    S.PARAMETERS_D.nElement())
                S.GRAD_PARAMETERS_D.add(OPT["D_L2"], S.PARAMETERS_D)
        S.CONFUSION.batchAdd(nn.as_plain(outputs), targets)
        if not OPT["fused_update"] and OPT["D_clamp"] != 0:
            S.GRAD_PARAMETERS_D.clamp(-OPT["D_clamp"], OPT["D_clamp"])
        S._last = dict(outputs_D=outputs.clone() if S.keep_outputs else outputs, f_D=f)
        if maxAccuracyD <= 1.0:  # the gate can trigger: read this batch's accuracy back (:115-166)
            o = nn.as_plain(outputs).numpy().reshape(-1)
            t = targets.numpy().reshape(-1)
            # every rank must take the same decision (a rank that skipped the update would desynchronise the
            # parameters): accuracy over the GLOBAL batch
            hits, tot = parallel.allreduce_sum_host([float(np.sum((o > 0.5) == (t > 0.5))), float(o.size)])
            tV = hits / tot
            S.accs.append(tV)
            if len(S.accs) > accsInterval:
                S.accs.pop(0)
            st["doTrainD"] = mean(S.accs) < maxAccuracyD
            if not st["doTrainD"]:
                return False, False
        return f, S.GRAD_PARAMETERS_D

    # --------------------------------------------------------- fevalG_on_D (adversarial.lua:171-215)
    def fevalG_on_D(x):
        if x is not S.PARAMETERS_G:
            S.PARAMETERS_G.copy(x)
        S.GRAD_PARAMETERS_G.zero()
        targets = buf["targets_G"]
        samples = st.pop("samples_pre", None)
        if st.pop("both", False):
            samples = S.MODEL_G.pairJoin()       # cg_net_pair_join: the step's stream waits for the second pass, which hands out its images
            off = st.pop("noise_off", None)
            if off is not None:
                r = rng()
                assert r.offset == off, "the G-step's noise was drawn from the wrong position of the counter stream"
                r.take(st["noiseInputs"].nElement())
        if samples is None:
            samples = nn_utils.createImagesFromNoise(S, st["noiseInputs"], False, True)
        outputs = S.MODEL_D.forward(samples)
        f = S.CRITERION.forward(outputs, targets)
        df_samples = S.CRITERION.backward(outputs, targets)
        if OPT["exact_reference_backward"]:
            S.MODEL_D.backward(samples, df_samples)
        else:
            S.MODEL_D.updateGradInput(samples, df_samples)
        df_do = S.MODEL_D.modules[0].gradInput
        if getattr(S.MODEL_G, "_bucket_overlap", False) and S.MODEL_G._planned_last:
            # the plan starts each gradient bucket's all-reduce as soon as its backward is complete (cg_net_set_dp): all but the
            # last bucket (Linear 100 -> 8192) travel under the remaining backward; nothing is ever one step stale
            S.MODEL_G.backward(st["noiseInputs"], df_do)
            if not S.MODEL_G._pnet[1].finish_buckets():       # no contiguous buckets: the whole vector at once
                parallel.allreduce_mean_(S.GRAD_PARAMETERS_G.t)
        else:
            S.MODEL_G.backward(st["noiseInputs"], df_do)
            parallel.allreduce_mean_(S.GRAD_PARAMETERS_G.t)
        if not OPT["fused_update"]:
            if OPT["G_L1"] != 0 or OPT["G_L2"] != 0:
                f = float(f) + OPT["G_L1"] * S.PARAMETERS_G.norm(1) + OPT["G_L2"] * S.PARAMETERS_G.norm(2) ** 2 / 2
                # adversarial.lua:206 scales the sign term by G_L2 (upstream slip, inert at the defaults)
                lib().axpy_sign(stream(), OPT["G_L2"], S.PARAMETERS_G.ptr, S.GRAD_PARAMETERS_G.ptr,
                                S.PARAMETERS_G.nElement())
                S.GRAD_PARAMETERS_G.add(OPT["G_L2"], S.PARAMETERS_G)
            if OPT["G_clamp"] != 0:
                S.GRAD_PARAMETERS_G.clamp(-OPT["G_clamp"], OPT["G_clamp"])
        S._last.update(outputs_G=outputs.clone() if S.keep_outputs else outputs, f_G=f, samples=samples)
        return f, S.GRAD_PARAMETERS_G

    # ----------------------------------------------------------------- (1) update D (:221-249)
    for _ in range(OPT["D_iterations"]):
        # (1.1) real data: N/2 random rows of the pool (math.random(trainData:size()), :226)
        if real_idx is None and S.device_rng:
            r = rng()
            lib().rng_randint_dev(stream(), buf["idx"].data_ptr(), half, trainData.size(), r.seed, r.take(half), r.base_ptr())
        else:
            idx = real_idx if real_idx is not None else S.random.randint(0, trainData.size(), size=half)
            buf["idx"].copy_(torch.from_numpy(np.asarray(idx, dtype=np.int32)), non_blocking=True)
        lib().gather_rows(stream(), trainData.pool.ptr, buf["idx"].data_ptr(), inputs.ptr, half, rowlen)
        # (1.2) sampled data
        noise = nn.to_device(noise_D) if noise_D is not None else nn_utils.stepNoiseInputs(S, half)
        both = _both_forwards_ok(S, N)
        if both:
            # ONE call below the ABI (cg_net_forward_pair, round 6): the fake-image pass on the step's stream, the G-step's pass on a library
            # stream of another hardware queue, its running statistics applied behind the first pass's.  The G-step's noise is drawn HERE at
            # the position it has in the reference's order: behind the D-step's dropout masks (drawn by MODEL_D:forward below)
            if noise_G is not None:
                st["noiseInputs"] = nn.to_device(noise_G)
            else:
                st["noise_off"] = rng().offset + S._d_draws[N]
                st["noiseInputs"] = nn_utils.stepNoiseInputsAt(S, N, st["noise_off"])
            samples = nn.as_nhwc(S.MODEL_G.forwardPair(noise, st["noiseInputs"]))
            st["both"] = True
        else:
            samples = nn.as_nhwc(nn_utils.createImagesFromNoise(S, noise, False))
        lib().memcpy_d2d(stream(), inputs.ptr + half * rowlen * 4, samples.ptr, half * rowlen * 4)
        S._last_fake = samples.clone() if S.keep_outputs else samples
        fused = dict(l1=OPT["D_L1"], l2=OPT["D_L2"], clamp=OPT["D_clamp"]) if OPT["fused_update"] else None
        m = OPT["D_optmethod"]  # adversarial.lua:240-248
        assert m in ("sgd", "adagrad", "adam"), "[Warning] Unknown optimizer method chosen for D."
        # Data-parallel overlap: D's gradient all-reduce travels over xGMI while the G-step's generator forward
        # (which reads no D parameter) runs; D's update is applied after the wait, before D sees those samples.
        st["overlap"] = (parallel.world_size() > 1 and OPT.get("overlap_comm", True) and OPT["fused_update"]
                         and OPT["D_iterations"] == 1 and OPT["G_iterations"] == 1)
        if st["overlap"]:
            fD, gD = fevalD(S.PARAMETERS_D)
            if "samples_pre" not in st and not st.get("both"):
                st["noiseInputs"] = nn.to_device(noise_G) if noise_G is not None else nn_utils.stepNoiseInputs(S, N)
                st["samples_pre"] = nn_utils.createImagesFromNoise(S, st["noiseInputs"], False, True)
            st.pop("pendingD").finish()
            getattr(optim, m)(lambda _x: (fD, gD), S.PARAMETERS_D, S.OPTSTATE[m]["D"], fused=fused)
        else:
            getattr(optim, m)(fevalD, S.PARAMETERS_D, S.OPTSTATE[m]["D"], fused=fused)
        if S.keep_outputs:   # tests: the gradient optim received (penalty + clamp applied), before the G-step touches D
            S._last["gD"] = S.GRAD_PARAMETERS_D.clone()

    # ----------------------------------------------------------------- (2) update G (:253-266)
    for _ in range(OPT["G_iterations"]):
        if "samples_pre" not in st and not st.get("both"):
            st["noiseInputs"] = nn.to_device(noise_G) if noise_G is not None else nn_utils.stepNoiseInputs(S, N)
        # upstream multiplies the L1 sign term by G_L2 (:206): keep that in the fused form too
        fused = dict(l1=OPT["G_L2"] if OPT["G_L1"] != 0 or OPT["G_L2"] != 0 else 0.0, l2=OPT["G_L2"],
                     clamp=OPT["G_clamp"]) if OPT["fused_update"] else None
        m = OPT["G_optmethod"]  # adversarial.lua:257-265
        assert m in ("sgd", "adagrad", "adam"), "[Warning] Unknown optimizer method chosen for G."
        getattr(optim, m)(fevalG_on_D, S.PARAMETERS_G, S.OPTSTATE[m]["G"], fused=fused)
        if S.keep_outputs:
            S._last["gG"] = S.GRAD_PARAMETERS_G.clone()
    return st["doTrainD"]


def _both_forwards_ok(S, N):
    """Both generator forwards of the iteration side by side (OPT.concurrent_g_both)?  One D and one G iteration, planned generator, and D's
    draw count at this batch known from an earlier iteration (the first iteration of a run goes one after the other).  Data parallel
    (round 6): allowed - the second pass's sync-BN all-reduces are ENQUEUED behind the first pass's on the same communicator, in the
    same order on every rank (host call order), so they cannot cross; the second pass then runs under D's forward / backward instead of
    beside the fake-image pass, which is still off the step's chain."""
    OPT = S.OPT
    G = S.MODEL_G
    last = G.modules[-1] if getattr(G, "modules", None) else None
    if isinstance(last, nn.Copy) and "Cuda" not in last.outtype:
        return False          # a net that hands host tensors back (models.lua:704 style wrapping) takes the plain path
    return bool(OPT.get("concurrent_g_both", False) and has_gpu() and nn.planned
                and OPT["D_iterations"] == 1 and OPT["G_iterations"] == 1 and N in S._d_draws
                and type(G) is nn.Sequential and getattr(G, "_pnet", None) and G._pnet[1] is not None and getattr(G, "_planned_last", False)
                and getattr(G._pnet[1], "last_draws", 0) == 0     # a G that draws (Dropout) would move the counter stream under the side pass
                and OPT["batchSize"] >= N)


class _Done:
    def finish(self):
        pass


def _bucketable(G):
    return isinstance(G, nn.Sequential) and type(G) is nn.Sequential and len(G.modules) > 1


class GraphedIteration:
    """The whole D+G iteration captured once into a hipGraph through the C ABI (cg_graph_begin / _end: every launch this host
    code makes between the two calls - on the capture stream and on the streams forked from it by events - is recorded) and
    replayed with cg_graph_launch: ~300 launches per step stop costing host time and inter-kernel gaps.  Everything that varies per
    step lives in device memory: the counter-stream position of every mask / noise / index draw (SplitMix.dev_base, advanced by
    `stride` draws per replay) and Adam's step counts (optim.adam device_step).  Only the inert-by-default accuracy gate
    (D_maxAcc <= 1) needs the host, so it is not available here.  Nothing may allocate during the capture: the eager warm-up
    passes compile every plan (cg_net_*: all buffers are allocated when a shape is first seen) and create every host-side buffer."""

    def __init__(self, S, trainData, thisBatchSize=None, warmup=2):
        import ctypes
        assert torch.cuda.is_available()
        self.S, self.data, self.N = S, trainData, thisBatchSize or S.OPT["batchSize"]
        S.device_rng = True
        for k in ("D", "G"):
            S.OPTSTATE["adam"][k]["device_step"] = True
        r = rng()
        self.base = r.enable_device_base()
        self.off0 = r.offset
        self.stride = None
        self.stream = torch.cuda.Stream()     # warm-up AND capture run here: per-stream scratch (column reductions, split-K
        self.stream.wait_stream(torch.cuda.current_stream())   # workspaces) must exist before the capture starts
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):  # eager passes: allocate every buffer, pack weights, learn the stride
                self._eager()
        torch.cuda.synchronize()
        ts = {k: S.OPTSTATE["adam"][k]["t"] for k in ("D", "G")}
        self.exec = ctypes.c_void_p()
        # The capture goes through hipStreamBeginCapture directly (not torch.cuda.graph, which would give it a private memory pool):
        # an allocation made while capturing would have its address baked into the graph while torch's caching allocator stays free
        # to hand the block to someone else.  The warm-up passes above exist so that nothing allocates here - and this is checked,
        # not assumed: torch's allocation counter and every plan's buffer count must not move across the capture.
        assert not getattr(S, "keep_outputs", False), "GraphedIteration: keep_outputs clones tensors inside the step; switch it off"
        before = self._alloc_marks()
        with torch.cuda.stream(self.stream):
            lib().graph_begin(stream())
            try:
                self._body()
            finally:
                lib().graph_end(stream(), ctypes.byref(self.exec))
        after = self._alloc_marks()
        if after != before:
            lib().graph_destroy(self.exec)
            self.exec = ctypes.c_void_p()
            raise RuntimeError(f"GraphedIteration: device memory was allocated while the iteration was being captured (torch "
                               f"allocations / plan buffers {before} -> {after}); the graph would replay on addresses the allocator "
                               f"may reuse.  Run more warm-up iterations or find the late allocation.")
        for k in ("D", "G"):   # capture launched nothing: the host step count must not move (it is what a checkpoint stores)
            S.OPTSTATE["adam"][k]["t"] = ts[k]
        self.replays = 0

    @staticmethod
    def _alloc_marks():
        """(allocations torch's caching allocator has served so far, buffers every live plan owns)"""
        from . import planned
        st = torch.cuda.memory_stats()
        nblocks = sum(len(pn.blocks) for pn in planned.LIVE)
        nbytes = sum(pn.stats()["bytes"] for pn in planned.LIVE)
        return (int(st.get("allocation.all.allocated", 0)), nblocks, int(nbytes))

    def _body(self):
        r = rng()
        r.offset = self.off0
        iteration(self.S, self.data, self.N)
        self.stride = r.offset - self.off0
        lib().counter_add(stream(), self.base.data_ptr(), self.stride)
        r.offset = self.off0   # between steps the stream position is r.offset + *base (what a checkpoint stores)

    def _eager(self):
        self._body()

    def __call__(self):
        lib().graph_launch(self.exec, stream())   # on the caller's current stream
        self.replays += 1
        for k in ("D", "G"):
            self.S.OPTSTATE["adam"][k]["t"] += 1

    def __del__(self):
        try:
            lib().graph_destroy(self.exec)
        except Exception:
            pass


def train(S, trainData, maxAccuracyD=1.01, accsInterval=20, verbose=True):
    """adversarial.train (adversarial.lua:27-292): one epoch."""
    OPT = S.OPT
    N_epoch = OPT["N_epoch"] if OPT["N_epoch"] > 0 else trainData.size()
    dataBatchSize = OPT["batchSize"] // 2
    t0 = time.time()
    countTrainedD = countNotTrainedD = 0
    if verbose:
        print(f"<trainer> Epoch #{S.EPOCH} [batchSize = {OPT['batchSize']}]")
    for t in range(1, N_epoch + 1, dataBatchSize):
        thisBatchSize = min(OPT["batchSize"], N_epoch - t + 1)
        if thisBatchSize < 4:
            if verbose:
                print(f"[INFO] skipping batch at t={t}, because its size is less than 4")
            break
        thisBatchSize -= thisBatchSize % 2
        if iteration(S, trainData, thisBatchSize, maxAccuracyD, accsInterval):
            countTrainedD += 1
        else:
            countNotTrainedD += 1
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.time() - t0
    tV = S.CONFUSION.updateValids()
    if verbose:
        print(f"<trainer> time required for this epoch = {dt:.0f} s")
        print(f"<trainer> time to learn 1 sample = {1000 * dt / N_epoch:f} ms")
        print(f"<trainer> trained D {countTrainedD} of {countTrainedD + countNotTrainedD} times.")
        print("Confusion of D:")
        print(S.CONFUSION)
    S.CONFUSION.zero()
    S.EPOCH += 1
    return tV
